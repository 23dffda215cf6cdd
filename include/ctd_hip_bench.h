/* ctd_hip_bench.h -- measurement hooks of libctd_hip.so, used by bench.py and tools/ only.
 *
 * NOT part of the drop-in boundary (include/ctd_hip.h): nothing in connecting_the_dots_amd/torchext calls these, and a
 * reference-side binding has no use for them.  They exist because SURVEY 8(d) asks for the dominant
 * kernel's launch duration measured with HIP events on the launch stream, and that kernel sits between two other
 * launches of one C call (pre-pass -> volume kernel -> fix-up).
 */
#ifndef CTD_HIP_BENCH_H
#define CTD_HIP_BENCH_H

#ifdef __cplusplus
extern "C" {
#endif

/* --------------------------------------------------------------------------------------
 * Per-kernel device timing for benchmarks (off by default, zero cost when off).
 * When enabled, every launch of the dominant kernel of ctd_xcorrvol_f32 /
 * ctd_xcorrvol_argmax_f32 (the NCC volume kernel proper, not its pre-pass) is bracketed by
 * a pair of hipEvents recorded on the launch stream.  ctd_kernel_timing_collect()
 * synchronises those events, returns the number of launches seen since the last collect
 * and their average duration in milliseconds, and releases the events.
 * `columns` receives the number of output columns per row that kernel covers (the
 * remaining W - columns are produced by a secondary kernel), so that the caller can
 * price the launch in algorithmic bytes.  The state is per calling THREAD: the thread that
 * enables it times its own launches; other threads are not instrumented and share nothing.
 * `enable` > 1 also says how many events to hold ready (two per launch until the next
 * collect; 128 by default): none is created or first recorded between two launches then.
 * -------------------------------------------------------------------------------------- */
void ctd_kernel_timing_enable(int enable);
int ctd_kernel_timing_collect(double* avg_ms, int* columns);

#ifdef __cplusplus
}
#endif

#endif /* CTD_HIP_BENCH_H */
