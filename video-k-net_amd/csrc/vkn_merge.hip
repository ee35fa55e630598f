// vkn_merge.hip — the THING-FIRST panoptic merge (`merge_joint=False`): KernelIterHead.merge_stuff_thing
// (knet/det/kernel_iter_head.py:385-465) and VideoKernelIterHead.merge_stuff_thing_thing_first (knet/video/kernel_iter_head.py:656-742).
//
// The reference pastes the K boolean full-resolution masks one by one in score order; every instance needs two whole-image sums
// (its area, its overlap with what is already painted) before the next one can be decided — three `.item()` host syncs per
// instance.  Here the loop stays on the device: per step two launches over the image,
//   k_mg_count  area and overlap of mask `k` (integer atomics: exact, order independent)
//   k_mg_paint  every workgroup re-derives the SAME accept / reject decision from the counters (fp64, as Python evaluates
//               `intersect * 1.0 / area > iou_thr`), paints the accepted pixels, and ONE thread records the segment and hands
//               the running state (next id, stop flag) to the next step through a per-step slot (no intra-launch race)
// and the host reads ONE table back at the end.  Stuff masks follow with their own rule (area of the still-empty part
// >= stuff_max_area).  HBM-bound on 1-byte masks + the int32 map; no matrix work.
#include "../../include/vkn.h"
#include "vkn_common.h"
#include "vkn_launch.h"

namespace {

struct MgState {       // one per step (slot s = state BEFORE step s)
    int next_id;       // current_segment_id so far
    int stop;          // things: a score below instance_score_thr was met -> skip the remaining things
};

__global__ __launch_bounds__(256) void k_mg_count(const unsigned char* __restrict__ masks, const int* __restrict__ order, int step,
                                                   const int* __restrict__ pan, int HW, unsigned* __restrict__ cnt) {
    const unsigned char* m = masks + (size_t)order[step] * HW;
    unsigned area = 0, inter = 0;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < HW; p += gridDim.x * 256) {
        const unsigned on = m[p] != 0;
        area += on;
        inter += on & (pan[p] > 0);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        area += __shfl_xor(area, o, 64);
        inter += __shfl_xor(inter, o, 64);
    }
    if ((threadIdx.x & 63) == 0 && (area | inter)) {
        atomicAdd(&cnt[2 * step + 0], area);
        atomicAdd(&cnt[2 * step + 1], inter);
    }
}

// kind 0: thing rule, kind 1: stuff rule.  info row of the step: [segment id | 0, kind, label, instance index | area, score bits]
__global__ __launch_bounds__(256) void k_mg_paint(const unsigned char* __restrict__ masks, const int* __restrict__ order, int step,
                                                   int kind, const float* __restrict__ scores, const int* __restrict__ labels,
                                                   int* __restrict__ pan, int HW, const unsigned* __restrict__ cnt,
                                                   MgState* __restrict__ st, int* __restrict__ info, double score_thr, double iou_thr,
                                                   int stuff_max_area) {
    const int k = order[step];
    const MgState s = st[step];
    const unsigned area = cnt[2 * step + 0], inter = cnt[2 * step + 1];
    bool accept = false, stop = s.stop != 0;
    unsigned final_area = area - inter;
    if (kind == 0) {
        if (!stop && (double)scores[k] < score_thr) stop = true;                           // `break`
        if (!stop && area > 0 && !((double)inter * 1.0 / (double)area > iou_thr) && final_area > 0) accept = true;
    } else {
        stop = false;                                                                      // the stuff loop has no break
        accept = (int)final_area >= stuff_max_area;
    }
    if (accept) {
        const unsigned char* m = masks + (size_t)k * HW;
        const int id = s.next_id + 1;
        for (int p = blockIdx.x * 256 + threadIdx.x; p < HW; p += gridDim.x * 256)
            if (m[p] != 0 && pan[p] == 0) pan[p] = id;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        st[step + 1] = MgState{s.next_id + (accept ? 1 : 0), stop ? 1 : 0};
        int* row = info + (size_t)step * 5;
        row[0] = accept ? s.next_id + 1 : 0;
        row[1] = kind;
        row[2] = labels[k];
        row[3] = kind == 0 ? k : (int)final_area;
        row[4] = kind == 0 ? __float_as_int(scores[k]) : 0;
    }
}

}  // namespace

extern "C" {

size_t vkn_merge_workspace_bytes(int Kt, int Ks) {
    if (Kt < 0 || Ks < 0) return 0;
    const size_t steps = (size_t)Kt + Ks;
    return ((steps * 2 * sizeof(unsigned) + 255) & ~(size_t)255) + (((steps + 1) * sizeof(MgState) + 255) & ~(size_t)255);
}

int vkn_panoptic_thing_first_u8(const unsigned char* thing_masks, const float* thing_scores, const int* thing_labels,
                                const int* thing_order, int Kt, const unsigned char* stuff_masks, const int* stuff_labels,
                                const int* stuff_order, int Ks, int HW, double instance_score_thr, double iou_thr,
                                int stuff_max_area, int* panoptic_seg, int* info, int* nseg, void* ws, size_t ws_bytes,
                                void* stream) {
    if (Kt < 0 || Ks < 0 || HW <= 0 || !panoptic_seg || !info || !nseg) return VKN_E_ARG;
    if (Kt > 0 && (!thing_masks || !thing_scores || !thing_labels || !thing_order)) return VKN_E_ARG;
    if (Ks > 0 && (!stuff_masks || !stuff_labels || !stuff_order)) return VKN_E_ARG;
    if (!ws || ws_bytes < vkn_merge_workspace_bytes(Kt, Ks)) return VKN_E_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t steps = (size_t)Kt + Ks;
    unsigned* cnt = static_cast<unsigned*>(ws);
    MgState* state = reinterpret_cast<MgState*>(static_cast<char*>(ws) + ((steps * 2 * sizeof(unsigned) + 255) & ~(size_t)255));
    if (hipMemsetAsync(ws, 0, vkn_merge_workspace_bytes(Kt, Ks), st) != hipSuccess) return VKN_E_LAUNCH;
    if (hipMemsetAsync(panoptic_seg, 0, (size_t)HW * sizeof(int), st) != hipSuccess) return VKN_E_LAUNCH;
    int blocks = (HW + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    for (int s = 0; s < Kt; ++s) {
        hipLaunchKernelGGL(k_mg_count, dim3(blocks), dim3(256), 0, st, thing_masks, thing_order, s, panoptic_seg, HW, cnt);
        hipLaunchKernelGGL(k_mg_paint, dim3(blocks), dim3(256), 0, st, thing_masks, thing_order, s, 0, thing_scores, thing_labels,
                           panoptic_seg, HW, cnt, state, info, instance_score_thr, iou_thr, stuff_max_area);
    }
    for (int s = 0; s < Ks; ++s) {
        // the stuff arrays are indexed from 0: shift the per-step slots by Kt through pointer offsets
        hipLaunchKernelGGL(k_mg_count, dim3(blocks), dim3(256), 0, st, stuff_masks, stuff_order, s, panoptic_seg, HW, cnt + 2 * Kt);
        hipLaunchKernelGGL(k_mg_paint, dim3(blocks), dim3(256), 0, st, stuff_masks, stuff_order, s, 1, (const float*)nullptr,
                           stuff_labels, panoptic_seg, HW, cnt + 2 * Kt, state + Kt, info + (size_t)5 * Kt, instance_score_thr,
                           iou_thr, stuff_max_area);
    }
    VKN_CHECK_LAUNCH();
    // number of segments = the running id after the last step
    if (hipMemcpyAsync(nseg, &state[steps].next_id, sizeof(int), hipMemcpyDeviceToDevice, st) != hipSuccess) return VKN_E_LAUNCH;
    return VKN_OK;
}

}  // extern "C"
