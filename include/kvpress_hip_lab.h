/* kvpress_hip_lab.h -- measurement and test aids of libkvpress_hip.so.
 *
 * NOT part of the boundary: include/kvpress_hip.h declares only entry points that replace lines of the reference (each cites
 * them) plus kvp_version / kvp_last_error / kvp_async_error_check.  What is declared here exists for bench.py (per-kernel
 * timing), the lab scripts under tools/ (clock probe, knob reload) and the residency tests of the one-launch select
 * (kvp_occupy_cus); a maintainer binding the library to the reference never needs this file. */
#ifndef KVPRESS_HIP_LAB_H
#define KVPRESS_HIP_LAB_H
#include "kvpress_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* ---- per-kernel timing -----------------------------------------------------------------------
 * kvp_prof_enable(1) makes every kernel launch of the calling thread record a HIP event pair on
 * its launch stream; after synchronising, kvp_prof_get(i) returns kernel i's name and duration.
 * kvp_prof_enable(0) turns it off and drops the records.  Used by bench.py for roofline.achieved. */
int kvp_prof_enable(int on);
int kvp_prof_count(void);
int kvp_prof_get(int i, const char** name, float* ms);
/* kvp_clock_probe: enqueue a one-wave kernel that spins spin_us microseconds and writes the shader clock (MHz, float, device
 * memory) it saw: s_memtime ticks per 100 MHz s_memrealtime tick.  Enqueued right behind a kernel it shows the clock that
 * kernel ran at (the governor is slow compared with a kernel). */
int kvp_clock_probe(float* mhz_out, int spin_us, kvp_stream_t stream);
/* kvp_occupy_cus (test aid): enqueue `blocks` workgroups of `threads` threads (a multiple of 64, <= 1024) that hold `lds_bytes`
 * of LDS each and spin spin_us microseconds (<= 200000) -- CUs that some other stream cannot use meanwhile.  The tests of the
 * one-launch select's residency behaviour run it beside kvp_topk_select (tests/test_gpu_cluster_failure.py). */
int kvp_occupy_cus(int blocks, int threads, int lds_bytes, int spin_us, kvp_stream_t stream);
/* kvp_tuning_reload: the library's tuning knobs (KVP_* environment variables: launch geometries and kernel-variant switches
 * for A/B runs) are read from the environment once, at first use, and cached; this drops the cache so that the next use of
 * every knob re-reads the environment.  Not needed in production; tests and lab scripts call it after changing a variable. */
int kvp_tuning_reload(void);

#ifdef __cplusplus
}
#endif
#endif /* KVPRESS_HIP_LAB_H */
