/*
 * effort_hip_debug.h -- profiling, tracing and A/B hooks of libeffort_hip.so.  NOT part of the drop-in boundary
 * (include/effort_hip.h): nothing in the reference corresponds to these; the bench, the tools/ scripts and a few tests
 * use them.  They may change between builds.
 */
#ifndef EFFORT_HIP_DEBUG_H
#define EFFORT_HIP_DEBUG_H

#include "effort_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Tuning knobs (nothing in the reference corresponds to them; the tests pin launch geometries with them). */
/* Override the launch geometry heuristics of the multiply kernel: waves per workgroup (4, 8 or 16),
 * elements per lane (1, 2 or 4) and number of row slices (0 = heuristic).  Returns EFFORT_ERR_ARG
 * for unsupported combinations. */
EFFORT_API int effort_set_tuning(effort_ctx* ctx, int wavesPerGroup, int elemsPerLane, int rowSlices);
/* split = 1: evaluate findCutoff32 in its own one-workgroup launch ahead of the multiply kernel instead of
 * inside it.  Costs a kernel boundary per call.  Results are bit-identical.  Default 0 (fused). */
EFFORT_API int effort_set_split_cutoff(effort_ctx* ctx, int split);

/* Group launches with more work items than wgPerCU workgroups per CU run as that many PERSISTENT workgroups pulling
 * items from per-XCD queues.  -1 = heuristic (default), 0 = always one workgroup per item. */
EFFORT_API int effort_set_persistent(effort_ctx* ctx, int wgPerCU);

/* Overlap mode (effort_set_overlap): the hooks effort_group_dispatch_count / effort_group_cutoff / effort_debug_slice_counts
 * read the lane the most recent launch went to; this points them at another lane's last launch (0 .. lanes-1). */
EFFORT_API int effort_debug_hook_lane(effort_ctx* ctx, int lane);

/* 1 in libeffort_hip_lab.so (built with -DEFFORT_LAB: device-clock stamps, per-item trace, ablation switches, environment knobs), 0 in the
 * shipped libeffort_hip.so, whose kernels carry none of that: there enable = 2 / 3 below, effort_kernel_clock, effort_debug_stamps and
 * effort_debug_trace return EFFORT_ERR_KIND, and enable = 1 records HIP events only. */
EFFORT_API int effort_is_lab_build(void);
/* Timing hooks.  enable = 1: HIP events are recorded on the context's stream around each launch (not capturable into a
 * graph) AND the multiply kernel stamps the device wall clock at its first workgroup's start / last workgroup's end;
 * enable = 2: device clock only (works inside hipGraph replays); 3: 2 plus a per-item trace (effort_debug_trace); 0: off.
 * effort_kernel_timing returns event-to-event averages in microseconds (they include the launch gap in front of each
 * kernel); effort_kernel_clock returns the multiply kernel's own average duration (first start -> last end).  Both
 * reset their accumulators. */
EFFORT_API int effort_enable_kernel_timing(effort_ctx* ctx, int enable);
EFFORT_API int effort_kernel_clock(effort_ctx* ctx, double* mul_us_avg, int* n_launches);
EFFORT_API int effort_kernel_timing(effort_ctx* ctx, double* mul_us_avg, double* cutoff_us_avg,
                         double* integrate_us_avg, int* n_samples);
/* resident workgroups per CU the runtime grants the (q4, waves, elems) multiply kernel at ldsBytes of LDS */
EFFORT_API int effort_debug_occupancy(effort_ctx* ctx, int q4, int waves, int elems, int ldsBytes);
/* 24 raw u64 phase stamps written by the most recent cutoff / multiply kernels in timing mode. */
EFFORT_API int effort_debug_stamps(effort_ctx* ctx, unsigned long long* host32);
/* Kept rows per row slice of call idx of the most recent (group) launch (their sum is dispatch.size); returns the
 * number of slices copied (<= maxSlices), or a negative error code. */
EFFORT_API int effort_debug_slice_counts(effort_ctx* ctx, int idx, uint32_t* host, int maxSlices);
/* enable = 3 (device clock + trace): every work item of the most recent multiply launch leaves a 64-byte record
 * {item | workgroup << 32 (bit 63: cutoff job), XCC_ID | HW_ID << 32, six device wall-clock stamps: start, staged,
 * cutoff, selected, streamed, handed over}; copies the first maxRecords (<= 4096) records to host (8 u64 each), followed
 * by 4 u64 per item: the device clock when wave 0 was a quarter, half and three quarters through its rows (host must
 * hold 12 * maxRecords u64). */
EFFORT_API int effort_debug_trace(effort_ctx* ctx, unsigned long long* host, int maxRecords);

#ifdef __cplusplus
}
#endif
#endif /* EFFORT_HIP_DEBUG_H */
