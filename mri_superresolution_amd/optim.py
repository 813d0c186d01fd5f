"""Fused Adam over the model's flat fp32 buffers (one HIP kernel per step).

Drop-in for ``optim.Adam(model.parameters(), lr, weight_decay)`` as used by
``/root/reference/scripts/train.py:186`` (L2 weight decay added to the gradient, eps 1e-8, betas
(0.9, 0.999), bias correction), and usable with ``optim.lr_scheduler.ReduceLROnPlateau``
(train.py:189-191) because it is a ``torch.optim.Optimizer`` with one param group.
``state_dict()`` is emitted in torch.optim.Adam's layout (per-parameter ``step`` / ``exp_avg`` /
``exp_avg_sq``) so checkpoints stay interchangeable (train.py:410-418).

fp16 autocast (train.py:303-311): ``scaler.step(optimizer)`` of ``torch.amp.GradScaler`` works unchanged - the class
advertises ``_step_supports_amp_scaling``, so the scaler hands over its ``grad_scale`` / ``found_inf`` DEVICE tensors
and the kernel un-scales, skips on overflow and counts steps on the device: no host synchronisation per step.
"""
from __future__ import annotations

import torch

from . import _lib as L


class FusedAdam(torch.optim.Optimizer):
    _step_supports_amp_scaling = True      # torch.amp.GradScaler.step: pass grad_scale / found_inf tensors, do not unscale

    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, device_step=False):
        if not hasattr(model, "flat_params"):
            raise TypeError("FusedAdam needs a model with flat parameter storage (UNetSuperRes)")
        self.model = model
        super().__init__(list(model.parameters()), dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._step = 0
        self._step_dev = None           # int32 device counter, used from the first GradScaler-driven step on
        self.device_step = bool(device_step)   # always count on the device (HIP-graph replay: graph_step.GraphedTrainStep)
        self.dp_grad_scale = 1.0        # 1/world_size when gradients were sum-all-reduced (data parallel)
        self._alloc()

    def _alloc(self):
        flat = self.model.flat_params
        self.exp_avg = torch.zeros_like(flat)
        self.exp_avg_sq = torch.zeros_like(flat)
        self._flat_ptr = flat.data_ptr()

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        m = self.model
        if m.flat_params.data_ptr() != self._flat_ptr:
            raise RuntimeError("model storage was re-created (model.to(...) after the optimizer was built)")
        if not m.flat_params.is_cuda:
            raise RuntimeError("FusedAdam runs on the GPU only (no CPU fallback)")
        plist = getattr(self, "_plist", None)
        if plist is None or self._plist_ptr != self._flat_ptr:
            # (parameter, its view of the flat gradient buffer): cached - named_parameters() walks the module tree
            plist = self._plist = [(p, m._grad_views[name]) for name, p in m.named_parameters()]
            self._plist_ptr = self._flat_ptr
        for p, gv in plist:
            if p.grad is not None and p.grad.data_ptr() != gv.data_ptr():
                gv.copy_(p.grad)                            # foreign gradient tensor: stage it
        g = self.param_groups[0]
        # torch.amp.GradScaler.step sets these two attributes (device tensors) around its call of step()
        loss_scale, found_inf = getattr(self, "grad_scale", None), getattr(self, "found_inf", None)
        if loss_scale is not None or found_inf is not None or self._step_dev is not None or self.device_step:
            if self._step_dev is None:      # first scaled step: the count moves to the device (skipped steps do not count)
                self._step_dev = torch.tensor([self._step], dtype=torch.int32, device=m.flat_params.device)
            ls = None if loss_scale is None else loss_scale.detach().to(torch.float32).reshape(-1)
            fi = None if found_inf is None else found_inf.detach().to(torch.float32).reshape(-1)
            L.call("mrisr_adam_step_amp", m.flat_params.data_ptr(), m.flat_grads.data_ptr(), self.exp_avg.data_ptr(),
                   self.exp_avg_sq.data_ptr(), m.flat_params.numel(), float(g["lr"]), float(g["betas"][0]),
                   float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]), self._step_dev.data_ptr(),
                   float(self.dp_grad_scale), L.ptr(ls), L.ptr(fi), L.stream_ptr(), nbytes=28 * m.flat_params.numel())
        else:
            self._step += 1
            L.call("mrisr_adam_step", m.flat_params.data_ptr(), m.flat_grads.data_ptr(), self.exp_avg.data_ptr(),
                   self.exp_avg_sq.data_ptr(), m.flat_params.numel(), float(g["lr"]), float(g["betas"][0]),
                   float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]), self._step, float(self.dp_grad_scale),
                   L.stream_ptr(), nbytes=28 * m.flat_params.numel())
        m.mark_weights_changed()        # the kernel wrote the masters through raw pointers: packed images are stale
        return loss

    # ---- torch.optim.Adam-compatible checkpoint format
    def _views(self, flat):
        out = []
        for name, p in self.model.named_parameters():
            off, n = self.model._offsets[name]
            v = flat[off:off + n]
            if p.dim() == 4:
                co, ci, kh, kw = p.shape
                v = v.view(co, kh, kw, ci).permute(0, 3, 1, 2)
            else:
                v = v.view(p.shape)
            out.append(v)
        return out

    @property
    def step_count(self) -> int:
        """Optimiser steps taken (reads the device counter back when a GradScaler drives the steps)."""
        return int(self._step_dev.item()) if self._step_dev is not None else self._step

    def set_step_count(self, step: int):
        self._step = int(step)
        if self._step_dev is not None:
            self._step_dev.fill_(int(step))

    def state_dict(self):
        groups = [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups]
        n = len(self.param_groups[0]["params"])
        groups[0]["params"] = list(range(n))
        state = {}
        step = self.step_count
        if step > 0:
            for i, (ea, es) in enumerate(zip(self._views(self.exp_avg), self._views(self.exp_avg_sq))):
                state[i] = {"step": torch.tensor(float(step)), "exp_avg": ea.clone(), "exp_avg_sq": es.clone()}
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd):
        for g, src in zip(self.param_groups, sd["param_groups"]):
            for k, v in src.items():
                if k != "params":
                    g[k] = v
        st = sd.get("state", {})
        if st:
            for i, (ea, es) in enumerate(zip(self._views(self.exp_avg), self._views(self.exp_avg_sq))):
                ent = st.get(i, st.get(str(i)))
                if ent is None:
                    continue
                ea.copy_(ent["exp_avg"])
                es.copy_(ent["exp_avg_sq"])
                self.set_step_count(int(float(ent["step"])))
