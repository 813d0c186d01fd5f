/*
 * mrisr.h — C-ABI of libmrisr.so, the MI355X (gfx950) kernel library behind the
 * U-Net super-resolution hot path of rdd0582/mri_superresolution.
 *
 * The reference has no FFI boundary of its own: its "operator API" is the Python
 * surface models/unet_model.py + utils/losses.py + the step loop of scripts/train.py,
 * which dispatch into torch/aten.  Each entry point below
 * replaces the aten ops one reference line (cited) dispatches; the Python mirror in
 * mri_superresolution_amd/ binds them through ctypes (INTEGRATION.md shows the stub).
 *
 * Conventions
 *  - ownership : every device buffer is allocated and owned by the caller (PyTorch);
 *                the library never allocates persistent memory.
 *  - errors    : every entry returns int (0 = ok, <0 = MRISR_E_*); mrisr_last_error()
 *                returns a thread-local message.
 *  - streams   : asynchronous launch on the caller's HIP stream (void* = hipStream_t),
 *                no internal synchronisation; re-entrant.  Process-wide state is limited to
 *                once-initialised constants (CU count, per-kernel LDS-size attributes, both
 *                behind std::call_once / magic statics); the library reads no environment variable.
 *  - layout    : activations NHWC ("channels last"), dtype MRISR_F32, MRISR_BF16 or MRISR_F16;
 *                conv weights [Cout][kh][kw][Cin] fp32 masters (= torch channels_last
 *                storage of a (Cout,Cin,kh,kw) tensor); statistics double / fp32.
 */
#ifndef MRISR_H
#define MRISR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MRISR_OK 0
#define MRISR_E_ARG (-1)
#define MRISR_E_SHAPE (-2)
#define MRISR_E_DTYPE (-3)
#define MRISR_E_HIP (-4)
#define MRISR_E_UNSUPPORTED (-5)

#define MRISR_F32 0
#define MRISR_BF16 1
#define MRISR_F16 2  /* IEEE half storage + f16 MFMA (fp32 accumulate): torch.amp.autocast's dtype, scripts/train.py:303-306 */

/* source transform applied while a convolution loads its input (never materialised) */
#define MRISR_SRC_RAW 0      /* value as stored                                              */
#define MRISR_SRC_NORM 1     /* LeakyReLU(0.2)(x*scale[n,c]+shift[n,c]) = GroupNorm+act fused */
#define MRISR_SRC_RELU 2     /* max(x,0) (VGG19 features)                                     */
#define MRISR_SP_NONE 0
#define MRISR_SP_POOL2 1     /* 2x2 max-pool of the transformed source  (unet_model.py:52)    */
#define MRISR_SP_UP2 2       /* bilinear x2, align_corners=True         (unet_model.py:71,151)*/
#define MRISR_COMBINE_CONCAT 0 /* torch.cat([src0, src1], 1)            (unet_model.py:93)    */
#define MRISR_COMBINE_BLEND 1  /* sigmoid(a)*src0 + (1-sigmoid(a))*src1 (unet_model.py:206-7) */
/* GroupNorm statistics buffers are [MRISR_STAT_SLOTS][N][groups][2] doubles (sum, sum of squares): producers
 * spread their atomics over the slots (same-address fp64 atomics serialise at ~170 ns each on gfx950),
 * mrisr_gn_finalize adds the slots up.  Zero the whole buffer before use.                                  */
#ifndef MRISR_STAT_SLOTS
#define MRISR_STAT_SLOTS 16
#endif
#define MRISR_SP_HEAD 3 /* mrisr_consumer.spatial only: see there */
#define MRISR_OUT_PLAIN 0
#define MRISR_OUT_PIXEL_SHUFFLE2 1 /* out[n,2y+i,2x+j,c/4] = conv[n,y,x,c], c=4c'+2i+j (unet_model.py:102) */

typedef struct {
    const void* ptr;     /* NHWC tensor [N][H][W][C] of the conv's dtype                       */
    const float* scale;  /* [N][C] (MRISR_SRC_NORM) or NULL                                    */
    const float* shift;  /* [N][C]                                                             */
    int32_t C, H, W;     /* stored dims of this source                                         */
    int32_t mode;        /* MRISR_SRC_*                                                        */
    int32_t spatial;     /* MRISR_SP_*                                                         */
    int32_t off_y, off_x;/* F.pad offsets of the source inside the conv input (unet_model.py:86-90) */
} mrisr_src;

typedef struct {
    int32_t dtype;          /* MRISR_F32 | MRISR_BF16 | MRISR_F16 (storage of src/out/packed weights) */
    int32_t N, H, W;        /* conv input (= output) spatial size                              */
    int32_t Cin, Cout;
    int32_t ksize;          /* 1 or 3 (padding ksize/2, stride 1)                              */
    int32_t nsrc;           /* 1 or 2                                                          */
    int32_t combine;        /* MRISR_COMBINE_*                                                 */
    int32_t out_mode;       /* MRISR_OUT_*                                                     */
    int32_t groups;         /* GroupNorm groups for the statistics epilogue (8), 0 = none      */
    int32_t relu_out;       /* 1: store max(y,0) (VGG)                                         */
    mrisr_src src[2];
    const float* blend_alpha; /* device scalar (pre-sigmoid), MRISR_COMBINE_BLEND only         */
    const void* wpacked;    /* from mrisr_pack_weights                                         */
    const float* bias;      /* [Cout] or NULL                                                  */
    void* out;              /* NHWC [N][H][W][Cout] (or pixel-shuffled [N][2H][2W][Cout/4])    */
    double* stats;          /* [MRISR_STAT_SLOTS][N][groups][2] (sum, sum of squares), accumulated; or NULL */
    const void* relu_mask;  /* NULL, or a tensor shaped like out: out is zeroed where relu_mask <= 0 (the ReLU
                               backward of the frozen VGG19 stack, utils/losses.py:95,145: this conv computes
                               dL/d(relu output), relu_mask = that relu output); 3x3, plain source/output only */
    int32_t cu_limit;       /* 0 = size the persistent grid for every CU of the device; k > 0: for k CUs only - for
                               launches that are meant to run BESIDE another stream's kernels (the weight gradients of
                               the backward pass next to the gradient chain) without either waiting for the other's CUs */
    int32_t reserved_;
    const void* wpacked_ring; /* NULL, or the same weight in the RING layout (mrisr_pack_weights with MRISR_PACK_RING set,
                               mrisr_packed_weight_bytes_ring > 0): mrisr_conv_forward then takes the deep-ring raw-source
                               kernel (csrc/conv_ring.hip) where the launch qualifies, and the classic image elsewhere  */
} mrisr_conv_desc;

const char* mrisr_last_error(void);
int mrisr_version(void);
/* the MRISR_STAT_SLOTS this library was compiled with (the caller sizes the statistics buffers with it) */
int mrisr_stat_slots(void);

/* ---- convolution: replaces nn.Conv2d forward (unet_model.py:29,34,72,101,152,168) and, with
 *      flipped/transposed packed weights and a RAW source, its input-gradient. ------------- */
/* bytes of the packed image for a (Cout,Cin,k,k) weight */
size_t mrisr_packed_weight_bytes(int dtype, int Cout, int Cin, int ksize);
/* w: fp32 [Cout][k][k][Cin].  transpose_flip bit 0 = 0: forward operand; 1: dgrad operand
 * (roles of Cin/Cout swapped, taps mirrored).  | MRISR_PACK_RING: the ring layout
 * [cout block of mrisr_conv_ring_bn()][cin chunk of 16][tap][row][32 B] read by csrc/conv_ring.hip
 * (packed buffer of mrisr_packed_weight_bytes_ring() bytes; Cout / Cin there are the OPERAND's, i.e. already
 * exchanged for the dgrad operand).                                                        */
#define MRISR_PACK_RING 256
/* output-channel block of the ring layout for an operand with these dims, 0 = the ring kernel does not take it */
int mrisr_conv_ring_bn(int dtype, int Cout, int Cin, int ksize);
size_t mrisr_packed_weight_bytes_ring(int dtype, int Cout, int Cin, int ksize);
int mrisr_pack_weights(int dtype, const float* w, int Cout, int Cin, int ksize, int transpose_flip,
                       void* packed, void* stream);
/* the same for many weights in ONE launch (a training step re-packs every layer after the optimiser update):
 * jobs_device = device array of njobs descriptors (pointers as in mrisr_pack_weights)                          */
typedef struct {
    const float* w;
    void* packed;
    int32_t Cout, Cin, ksize, transpose_flip;
} mrisr_pack_job;
int mrisr_pack_weights_batched(int dtype, const mrisr_pack_job* jobs_device, int njobs, void* stream);
int mrisr_conv_forward(const mrisr_conv_desc* d, void* stream);
/* writes the name of the kernel instantiation mrisr_conv_forward (wgrad=0) / mrisr_conv_wgrad (wgrad=1) will
 * launch for this descriptor, template arguments as in the mangled symbol rocprofv3 reports               */
int mrisr_conv_variant(const mrisr_conv_desc* d, int wgrad, char* out, size_t n);
/* weight gradient: dw[Cout][k][k][Cin] (fp32, ACCUMULATED) = sum_pix dy[pix][co] * in[pix+tap][ci];
 * the input is described exactly as in the forward desc (d->out, d->wpacked, d->bias ignored). */
int mrisr_conv_wgrad(const mrisr_conv_desc* d, const void* dy, float* dw, float* workspace, size_t workspace_floats,
                     void* stream);
/* workspace (optional, caller-owned scratch of >= mrisr_conv_wgrad_workspace_floats(d) floats): the split-K partial
 * sums go through it and a second kernel adds them into dw; with NULL (or a smaller buffer) they are added with
 * float atomics directly (same result up to summation order, slower).                                            */
size_t mrisr_conv_wgrad_workspace_floats(const mrisr_conv_desc* d);

/* stem conv Cin==1 (unet_model.py:29 for "inc"): x fp32 [N][H][W], w fp32 [Cout][9]          */
int mrisr_stem_forward(int dtype, const float* x, const float* w, void* out, double* stats,
                       int N, int H, int W, int Cout, int groups, void* stream);
int mrisr_stem_wgrad(int dtype, const float* x, const void* dy, float* dw,
                     int N, int H, int W, int Cout, void* stream);
/* the same for 1 <= Cin <= 4 image channels (UNetSuperRes(in_channels=...), unet_model.py:129,137): x fp32 NCHW
 * [N][Cin][H][W], w / dw fp32 [Cout][9][Cin] (the channels-last storage every conv weight has on this side)  */
int mrisr_stem_forward_multi(int dtype, const float* x, const float* w, void* out, double* stats,
                             int N, int H, int W, int Cin, int Cout, int groups, void* stream);
int mrisr_stem_wgrad_multi(int dtype, const float* x, const void* dy, float* dw,
                           int N, int H, int W, int Cin, int Cout, void* stream);

/* ---- GroupNorm(8,C)+LeakyReLU(0.2): statistics -> per-(n,c) affine (unet_model.py:30-31) -- */
/* stats [MRISR_STAT_SLOTS][N][G][2] double -> scale/shift [N][C] fp32, meanrstd [N][G][2] fp32; count = (C/G)*H*W */
int mrisr_gn_finalize(const double* stats, const float* gamma, const float* beta, float* scale,
                      float* shift, float* meanrstd, int N, int C, int groups, double count,
                      float eps, void* stream);

/* out [N][H/2][W/2][C] = MaxPool2d(2)(LeakyReLU(x*scale+shift))  (unet_model.py:52): materialised pooled
 * activation, so the encoder convolutions read a plain tensor.                                            */
int mrisr_norm_pool2(int dtype, const void* x, const float* scale, const float* shift, void* out, int N,
                     int H, int W, int C, void* stream);
/* out [N][2h][2w][C] = bilinear x2 (align_corners=True) of LeakyReLU(x*scale+shift): materialised input of
 * final_up_bilinear's 3x3 conv (unet_model.py:151-152).                                                     */
int mrisr_norm_upsample2(int dtype, const void* x, const float* scale, const float* shift, void* out, int N,
                         int h, int w, int C, void* stream);
/* out [N][H][W][C] = sigmoid(alpha)*LeakyReLU(x0*scale0+shift0) + (1-sigmoid(alpha))*LeakyReLU(x1*scale1+shift1): the
 * materialised alpha blend of the two head branches (unet_model.py:206-207), input of final_conv.0.               */
int mrisr_norm_blend(int dtype, const void* x0, const float* scale0, const float* shift0, const void* x1,
                     const float* scale1, const float* shift1, const float* alpha, void* out, int N, int H, int W,
                     int C, void* stream);
/* z [N][2h][2w][C] = bilinear x2 (align_corners=True) of z_low [N][h][w][C], plus GroupNorm statistics of z
 * (stats [N][groups][2] double, accumulated; may be NULL).  With mrisr_conv_forward on the low-resolution
 * tensor this evaluates nn.Upsample -> nn.Conv2d(1x1) (unet_model.py:71-72) as conv -> upsample.           */
int mrisr_upsample2_stats(int dtype, const void* z_low, void* z, double* stats, int N, int h, int w, int C,
                          int groups, void* stream);
/* The same pair in ONE launch for 16-bit storage: z [N][2h][2w][Cout] = bilinear x2 of conv1x1(LeakyReLU(x*scale+shift)) and
 * the GroupNorm statistics of z, the low-resolution tensor staying in LDS (csrc/up_fused.hip).  x [N][h][w][Cin] raw,
 * scale / shift [N][Cin], wpacked = mrisr_pack_weights image of the (Cout,Cin,1,1) weight; Cin % 32 == 0, Cout % 64 == 0.   */
int mrisr_up_conv1x1_fused(int dtype, const void* x, const float* scale, const float* shift, const void* wpacked, void* z,
                           double* stats, int N, int h, int w, int Cin, int Cout, int groups, void* stream);
/* adjoint of the above interpolation: dz [N][2h][2w][C] -> dz_low [N][h][w][C]                              */
int mrisr_upsample2_adjoint(int dtype, const void* dz, void* dz_low, int N, int h, int w, int C, void* stream);

/* consumer of an activation in the backward pass */
typedef struct {
    const void* da;       /* NHWC gradient w.r.t. the consumer conv's (virtual) input           */
    int32_t C_total;      /* channel stride of da                                               */
    int32_t c_off;        /* first channel of this producer inside da                           */
    int32_t H, W;         /* spatial dims of da                                                 */
    int32_t spatial;      /* MRISR_SP_* that the consumer applied to this producer              */
    int32_t off_y, off_x; /* pad offsets (MRISR_SP_NONE consumers)                              */
    int32_t weight_mode;  /* 0: plain; 1: times sigmoid(alpha); 2: times 1-sigmoid(alpha)       */
    /* MRISR_SP_HEAD only (NULL otherwise): the consumer is the output head nn.Conv2d(C, 1, 1) + sigmoid
     * (unet_model.py:172, 211) and dL/dact is never materialised: da = dL/dout [N][H][W] fp32, head_out = the
     * sigmoid output, head_w [C]:  dL/dact[n,y,x,c] = da*out*(1-out) * head_w[c].  The head's own gradients
     * (mrisr_head_backward's job) come out of the same two passes: mrisr_act_bwd_reduce accumulates per-image
     * partial sums into head_part [N][C+1] (zeroed scratch: sum dz*act per channel, then sum dz), and
     * mrisr_act_bwd_apply_fused (with fin) adds them to head_dw [C] and head_db [1].                              */
    const float* head_out;
    const float* head_w;
    float* head_part;
    float* head_dw;
    float* head_db;
} mrisr_consumer;

/* backward of LeakyReLU+GroupNorm for one producer tensor x [N][H][W][C] (raw conv output); restates
 * aten leaky_relu_backward + native_group_norm_backward + the adjoints of max_pool2d / upsample_bilinear2d
 * / cat / blend that autograd runs for unet_model.py:30-31,52,71,93,206-207.
 *  reduce  : gathers dL/dact from up to 2 consumers, applies LeakyReLU', writes g = dL/d(gn out) (dtype; g may be
 *            NULL when mrisr_act_bwd_apply_fused is used for pass 2)
 *            and accumulates red[N][C][2] += (sum g, sum g*xhat) (fp32)
 *  finalize: dgamma[C] += sum_n red[..][1], dbeta[C] += sum_n red[..][0], coef[3][N][C] such that
 *            dx = g*coef0 + x*coef1 + coef2      (count = (C/groups)*H*W)
 *  apply   : writes dx (dtype); out_mode PIXEL_SHUFFLE2 stores it un-shuffled as [N][H/2][W/2][4C]; dbias (optional,
 *            that mode only, [4C] fp32 accumulated) += per-channel sums of dx = the producing conv's bias gradient. */
/* alpha_slots (optional, 256 zeroed floats; first consumer plain, blend_alpha set): receives partial sums of
 * (first consumer's unweighted gradient) * activation; mrisr_act_bwd_finalize turns them into
 * dalpha += alpha_sign * sigmoid'(alpha) * sum  (+1 for the sigmoid(alpha) branch, -1 for the other; the two branches'
 * terms add up to unet_model.py:206-207's dL/dalpha, so mrisr_blend_alpha_grad's extra pass is not needed).      */
int mrisr_act_bwd_reduce(int dtype, const void* x, const float* scale, const float* shift,
                         const float* meanrstd, int nconsumers, const mrisr_consumer* consumers,
                         const float* blend_alpha, void* g, float* red, float* alpha_slots, int N, int H, int W,
                         int C, int groups, void* stream);
int mrisr_act_bwd_finalize(const float* red, const float* gamma, const float* meanrstd, float* dgamma,
                           float* dbeta, float* coef, int N, int C, int groups, double count,
                           const float* alpha_slots, const float* alpha, float* dalpha, float alpha_sign, void* stream);
int mrisr_act_bwd_apply(int dtype, const void* x, const void* g, const float* coef, void* dx, int N,
                        int H, int W, int C, int out_mode, float* dbias, void* stream);
/* apply without the intermediate tensor: when every consumer is MRISR_SP_NONE, mrisr_act_bwd_reduce may be called
 * with g = NULL and this entry gathers dL/dact from the consumers again (same arguments as the reduce pass).
 * Exactly one of coef / fin: with fin the finalize step runs inside this launch (the coefficients are derived from
 * red by every workgroup, dgamma / dbeta / dalpha are accumulated once) and mrisr_act_bwd_finalize is not called.   */
typedef struct mrisr_gn_bwd_fin {
    const float* red;          /* [N][C][2] of the finished reduce pass                                   */
    const float* gamma;        /* [C]                                                                     */
    const float* meanrstd;     /* [N][groups][2]                                                          */
    float* dgamma;             /* [C] accumulated                                                         */
    float* dbeta;              /* [C] accumulated                                                         */
    const float* alpha_slots;  /* optional, as for mrisr_act_bwd_finalize                                 */
    const float* alpha;
    float* dalpha;
    double count;              /* (C/groups)*H*W                                                          */
    float alpha_sign;
    int32_t groups;            /* <= 32                                                                   */
} mrisr_gn_bwd_fin;
int mrisr_act_bwd_apply_fused(int dtype, const void* x, const float* scale, const float* shift, int nconsumers,
                              const mrisr_consumer* consumers, const float* blend_alpha, const float* coef,
                              const mrisr_gn_bwd_fin* fin, void* dx, int N, int H, int W, int C, void* stream);
/* ONE-PASS form of mrisr_act_bwd_reduce + mrisr_act_bwd_apply_fused (csrc/norm.hip: act_bwd_onepass_kernel) for nodes whose
 * consumers are all plain (MRISR_SP_NONE), have the node's geometry and no blend weight; 16-bit storage.  x and every
 * consumer gradient are read from HBM once instead of twice: the blocks of an image keep their operands in registers across
 * an in-kernel image barrier (`arrive`: [N][mrisr_act_bwd_onepass_barrier_words()] uint32).  red ([mrisr_act_bwd_onepass_slots()][N][C][2] fp32, = fin->red:
 * the per-(n,c) sums spread over 16 copies) and arrive must be ZERO on entry.  mrisr_act_bwd_onepass_ok() != 0 tells whether a node qualifies (channel count, blocks per image <= 256: one
 * image's blocks must be resident together); the entry point refuses the others with MRISR_E_UNSUPPORTED.
 * Replaces the GroupNorm + LeakyReLU part of autograd's backward of DoubleConv (/root/reference/models/unet_model.py:28-37). */
int mrisr_act_bwd_onepass_slots(void);
int mrisr_act_bwd_onepass_barrier_words(void);
int mrisr_act_bwd_onepass_ok(int dtype, int nconsumers, const mrisr_consumer* consumers, int N, int H, int W, int C);
int mrisr_act_bwd_onepass(int dtype, const void* x, const float* scale, const float* shift, const float* meanrstd,
                          int nconsumers, const mrisr_consumer* consumers, float* red, uint32_t* arrive,
                          const mrisr_gn_bwd_fin* fin, void* dx, int N, int H, int W, int C, void* stream);
/* the same for a pixel-shuffled node (unet_model.py:102): one plain consumer with the node's geometry, even H and W; dx
 * is stored un-shuffled as [N][H/2][W/2][4C] (channel 4c + 2(Y&1) + (X&1)) and dbias (optional, [4C] fp32 accumulated) +=
 * the channel sums of dx = the producing conv's bias gradient (as mrisr_act_bwd_apply's PIXEL_SHUFFLE2 mode).       */
int mrisr_act_bwd_apply_fused_unshuffle(int dtype, const void* x, const float* scale, const float* shift,
                                        const mrisr_consumer* consumer, const float* blend_alpha,
                                        const mrisr_gn_bwd_fin* fin, void* dx, float* dbias, int N, int H, int W, int C,
                                        void* stream);
/* out[C] += sum over pixels of x[npix][C]  (bias gradient of nn.Conv2d(bias=True), unet_model.py:101) */
int mrisr_channel_sum(int dtype, const void* x, float* out, size_t npix, int C, void* stream);
/* dalpha += sigmoid'(alpha) * sum da * (act0 - act1)   (unet_model.py:206-207)               */
int mrisr_blend_alpha_grad(int dtype, const void* da, const void* x0, const float* scale0,
                           const float* shift0, const void* x1, const float* scale1,
                           const float* shift1, const float* alpha, float* dalpha, int N, int H,
                           int W, int C, void* stream);

/* ---- output head: GN+LReLU -> conv1x1(C->1)+bias -> sigmoid (unet_model.py:172,211) -------- */
int mrisr_head_forward(int dtype, const void* x, const float* scale, const float* shift,
                       const float* w, const float* b, float* out, int N, int H, int W, int C,
                       void* stream);
/* dout [N][H][W] fp32 -> da [N][H][W][C] (dtype), dw[C] and db (fp32, accumulated)           */
int mrisr_head_backward(int dtype, const void* x, const float* scale, const float* shift,
                        const float* w, const float* out, const float* dout, void* da, float* dw,
                        float* db, int N, int H, int W, int C, void* stream);
/* the same for 1 <= K <= 4 output channels (UNetSuperRes(out_channels=...), unet_model.py:129,172): w [K][C], b [K],
 * out / dout fp32 NCHW [N][K][H][W]; dw [K][C], db [K]                                                        */
int mrisr_head_forward_multi(int dtype, const void* x, const float* scale, const float* shift,
                             const float* w, const float* b, float* out, int N, int H, int W, int C,
                             int K, void* stream);
int mrisr_head_backward_multi(int dtype, const void* x, const float* scale, const float* shift,
                              const float* w, const float* out, const float* dout, void* da,
                              float* dw, float* db, int N, int H, int W, int C, int K, void* stream);

/* ---- loss: fused L1 + Gaussian-window SSIM (utils/losses.py:27-81,200-226) ---------------- */
/* a, b: [N][H][W] fp32 (single channel), 11-tap window.  sums[N][2] (double, accumulated):
 * sums[n][0] += sum|a-b|, sums[n][1] += sum ssim_map.  coef (optional, [3][N][H][W] fp32) receives
 * dS/dmu1, dS/dE[x^2], dS/dE[xy] for the backward pass.                                        */
int mrisr_ssim_l1_forward(const float* a, const float* b, double* sums, float* coef, int N, int H,
                          int W, float val_range, float sigma, void* stream);
/* da = gscale[0] * ( l1_w * sign(a-b) - ssim_w * [0<=mean ssim<=1] * d(sum ssim)/da ) / numel
 * (gscale: device scalar, NULL = 1; sums NULL = no clamp mask, for the bare ssim() metric)       */
int mrisr_ssim_l1_backward(const float* a, const float* b, const float* coef, const double* sums,
                           const float* gscale, float l1_w, float ssim_w, float* da, int N, int H,
                           int W, float sigma, void* stream);
/* the same two with the window size of utils/losses.py:27 (`window_size`: odd, 3 .. 15; the entries above are these with 11).
 * The gradient w.r.t. the SECOND image (SSIM and L1 are symmetric): swap a and b in both calls.                           */
int mrisr_ssim_l1_forward_win(const float* a, const float* b, double* sums, float* coef, int N, int H, int W,
                              float val_range, float sigma, int window_size, void* stream);
int mrisr_ssim_l1_backward_win(const float* a, const float* b, const float* coef, const double* sums,
                               const float* gscale, float l1_w, float ssim_w, float* da, int N, int H, int W,
                               float sigma, int window_size, void* stream);

/* CombinedLoss scalar without the perceptual term (utils/losses.py:200-226): out[0] = l1_w*L1 +
 * ssim_w*(1-clamp(SSIM,0,1)), out[1] = L1 mean, out[2] = SSIM mean, out[3+n] = per-sample SSIM.   */
int mrisr_loss_finalize(const double* sums, int N, int H, int W, float l1_w, float ssim_w, float* out,
                        void* stream);

/* ---- perceptual loss: glue of the frozen VGG19 feature stack (utils/losses.py:83-151).  The 3x3 convolutions are
 *      mrisr_conv_forward calls (bias + relu_out forward; relu_mask for the input gradient). ------------------- */
/* channels of the normalised VGG input tensor (3 real + zero padding to one 16-byte bf16 vector) */
int mrisr_vgg_input_channels(void);
/* x [npix] fp32 gray -> out [npix][8] (dtype): repeat(1,3,1,1) + (x-mean)/std (losses.py:105-114), channels 3..7 = 0 */
int mrisr_vgg_input_forward(int dtype, const float* x, void* out, size_t npix, void* stream);
/* adjoint: dimg[npix] (fp32, ACCUMULATED) += gscale[0]*scale * sum_c dx3[npix][c]/std_c  (gscale: device scalar or NULL) */
int mrisr_vgg_input_backward(int dtype, const void* dx3, const float* gscale, float scale, float* dimg,
                             size_t npix, void* stream);
/* nn.MaxPool2d(2) on NHWC: out [N][H/2][W/2][C] */
int mrisr_maxpool2_forward(int dtype, const void* x, void* out, int N, int H, int W, int C, void* stream);
/* its adjoint (first maximum wins, as aten): dx [N][H][W][C] from dy [N][H/2][W/2][C]; relu_gate=1 also zeroes dx
 * where x <= 0 (x is then the output of the nn.ReLU in front of the pool: ReLU backward fused)                  */
int mrisr_maxpool2_backward(int dtype, const void* x, const void* dy, void* dx, int N, int H, int W, int C,
                            int relu_gate, void* stream);
/* nn.L1Loss (kind 0) / nn.MSELoss (kind 1) between two feature tensors of n elements (losses.py:127-131,150):
 * out[0] = mean; sum16 = 16 doubles of scratch; da (optional, dtype) receives the UNSCALED gradient sign(a-b) or
 * 2(a-b) (times [a>0] when relu_gate=1: a is a ReLU output) - the 1/n and the upstream gradient are applied by
 * mrisr_vgg_input_backward at the end of the (linear) input-gradient chain.                                      */
int mrisr_feature_loss(int dtype, const void* a, const void* b, size_t n, int kind, double* sum16, float* out,
                       void* da, int relu_gate, void* stream);

/* ---- optimiser: torch.optim.Adam with L2-coupled weight decay (scripts/train.py:186) ------- */
/* grad_scale multiplies g first (1/world_size after a sum all-reduce).                        */
int mrisr_adam_step(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1,
                    float beta2, float eps, float weight_decay, int step, float grad_scale,
                    void* stream);
/* The same update under torch.amp.GradScaler's device-side contract (fp16 autocast, scripts/train.py:303-311):
 * gradients are multiplied by grad_mul / *loss_scale_device, the update is skipped when *found_inf_device != 0, and
 * the bias-correction step count *step_device (int32 on the device, starts at 0) advances only when the update ran.
 * loss_scale_device / found_inf_device may be NULL (= 1 / never).  No host read-back.                              */
int mrisr_adam_step_amp(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2,
                        float eps, float weight_decay, int* step_device, float grad_mul,
                        const float* loss_scale_device, const float* found_inf_device, void* stream);

/* ---- layout helpers ----------------------------------------------------------------------- */
int mrisr_cast(int src_dtype, const void* src, int dst_dtype, void* dst, size_t n, void* stream);

/* ---- inference pre / post-processing on the device (scripts/infer.py:97-130, 276, 331) ---- */
/* hist[batch][256] += histogram of batch 8-bit images of pixels_per_image bytes each (zero it first).               */
int mrisr_u8_histogram(const uint8_t* img, size_t pixels_per_image, int batch, unsigned* hist, void* stream);
/* out[b][i] = (clip(img[b][i], lo_b, hi_b) - lo_b) / (hi_b - lo_b) with lo_b / hi_b = np.percentile(img[b], q_lo / q_hi)
 * (method 'linear', numpy's float32 arithmetic); unnormalised clipped values when hi_b <= lo_b (infer.py:115-117).
 * lohi (optional) receives [batch][2] = (lo_b, hi_b).                                                                */
int mrisr_u8_percentile_normalise(const uint8_t* img, const unsigned* hist, size_t pixels_per_image, int batch,
                                  double q_lo, double q_hi, float* out, float* lohi, void* stream);
/* out[i] = (uint8)(clamp(x[i], 0, 1) * 255), truncating like ndarray.astype(np.uint8) (infer.py:276, 331).          */
int mrisr_f32_to_u8(const float* x, uint8_t* out, size_t n, void* stream);

/* ---- paired augmentation on the device (utils/dataset.py:138-175) ------------------------- */
typedef struct {
    float cos_a, sin_a;   /* of PIL's inverse rotation angle -radians(angle_degrees)                                  */
    int32_t rotate;       /* 0: no rotation                                                                            */
    int32_t flip;         /* horizontal flip first (TF.hflip), then the rotation                                       */
    int32_t fill;         /* rotation fill value = int(mean of the un-augmented image) (dataset.py:152-155)            */
    float brightness;     /* ImageEnhance.Brightness factor, 1 = none                                                  */
} mrisr_aug_geo;
typedef struct {
    float contrast;       /* ImageEnhance.Contrast factor, 1 = none                                                    */
    int32_t mean;         /* int(mean + 0.5) of the image entering the contrast stage                                  */
    float noise_sigma;    /* Gaussian noise in uint8 units (dataset.py:169-172: noise_std * 255), 0 = none             */
    uint32_t seed;        /* per-sample seed of the counter-based noise generator                                      */
} mrisr_aug_photo;
/* out[b] = brightness(rotate(flip(in[b]))) as uint8, PIL's NEAREST / truncation semantics; params: batch structs on the
 * device.  mean_device (optional, [batch] doubles): per-image means kept on the device - fill = int(mean[b]) then
 * overrides params[b].fill, so that no statistic has to be read back by the host.                                     */
int mrisr_augment_geo_u8(const uint8_t* in, uint8_t* out, int batch, int H, int W, const mrisr_aug_geo* params_device,
                         const double* mean_device, void* stream);
/* out[b] = ToTensor(noise(contrast(in[b]))) as fp32 in [0,1]; mean_device (optional): mean = int(mean[b] + 0.5).      */
int mrisr_augment_finish_u8(const uint8_t* in, float* out, int batch, size_t pixels_per_image,
                            const mrisr_aug_photo* params_device, const double* mean_device, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MRISR_H */
