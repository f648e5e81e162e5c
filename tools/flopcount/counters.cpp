long long fbo_cnt[6] = {0, 0, 0, 0, 0, 0};
extern "C" void fbo_flop_counters(long long* out, int reset) { for (int k = 0; k < 6; k++) { out[k] = fbo_cnt[k]; if (reset) fbo_cnt[k] = 0; } }
