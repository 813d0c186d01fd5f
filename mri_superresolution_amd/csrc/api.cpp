#include "common.h"
thread_local char g_mrisr_err[512] = "";
extern "C" const char* mrisr_last_error(void) { return g_mrisr_err; }
extern "C" int mrisr_version(void) { return 304; }   // 304: mrisr_{stem,head}_*_multi; 303: mrisr_ssim_l1_{forward,backward}_win; 302: mrisr_act_bwd_onepass(_ok); 301: mrisr_up_conv1x1_fused; 300: mrisr_conv_desc.wpacked_ring, ring weight layout (MRISR_PACK_RING), mrisr_conv_ring_bn
extern "C" int mrisr_stat_slots(void) { return MRISR_STAT_SLOTS; }
