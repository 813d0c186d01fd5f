#include "common.h"
thread_local char g_mrisr_err[512] = "";
extern "C" const char* mrisr_last_error(void) { return g_mrisr_err; }
extern "C" int mrisr_version(void) { return 210; }   // 210: mrisr_consumer grew the head fields, mrisr_gn_bwd_fin, apply_fused(fin), apply_fused_unshuffle
extern "C" int mrisr_stat_slots(void) { return MRISR_STAT_SLOTS; }
