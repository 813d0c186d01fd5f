export MRISR_LIB=$PWD/mri_superresolution_amd/libmrisr_tune.so
for t in 0 1 0 1; do echo "== TILE16=$t"; MRISR_TILE16=$t timeout -k 10 120 python tools/conv_bench.py --kinds fwd,dgrad --filter "fin.c0,fin.bil,up3.c3,down1.3,down2.3" --iters 20 --raw 2>/dev/null | grep -v "^total"; done
