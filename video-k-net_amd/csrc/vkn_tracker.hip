// vkn_tracker.hip — quasi-dense embedding association of the video models on the device: what
// `QuasiDenseEmbedTracker.match` + `update_memo` + `memo` decide per frame (knet/video/qdtrack/trackers/quasi_dense_embed_tracker.py:47-207),
// as ONE single-workgroup kernel per frame over a device-resident memo.
//
// Why one workgroup: the association is order dependent by definition — detections are visited in score order and a matched
// track is taken away from every later detection — and tiny (n <= 256 detections against m <= a few hundred memo entries of
// E <= 256 floats).  The reference walks it with Python loops over host tensors (one `.item()`-style sync per decision); here
// the inputs (thing boxes from vkn_panoptic_joint_f32, tracking embeddings from the head) are already on the device and never
// leave it: the kernel reads them, keeps the memo in a caller-owned state buffer, and returns the compacted detections with
// their ids.  The host reads back ONE int (the number of surviving detections) — or nothing with the padded API.
//
// Data structures (all in the caller's `state`, laid out by trk_carve):
//   tracklet table, structure of arrays, rows kept in CREATION order (= the iteration order of the reference's dict, which is
//   the column order of its score matrix and therefore the arg-max tie order): id, label, last_frame, acc_frame, box[5], velocity[5],
//   embed[E].  Expired rows are squeezed out by a stable compaction.
//   backdrop frames [F][D]: unmatched low-score detections of the last `memo_backdrop_frames` frames, newest first.
// Phases of the kernel (one __syncthreads between them; 1024 threads):
//   A  stable rank sort by score (descending)                 B  duplicate suppression: IoU against EVERY better-scored box
//   C  compaction of the survivors + outputs                  D  similarity [nv x m] (fp64 accumulation), bi-softmax / softmax / cosine
//   E  greedy assignment on ONE wave (running arg-max over the not-yet-taken columns, first index on ties)
//   F  new ids by prefix count                                 G  memo update: momentum embeddings, velocities, births, backdrops, expiry
// Floating-point contraction is OFF in this file: every value that feeds a decision is computed with the same sequence of
// individually rounded fp32 operations as the reference's element-wise tensor ops (IoU, momentum update, softmax division).
#include "../../include/vkn.h"
#include "vkn_common.h"

#pragma clang fp contract(off)

namespace {

constexpr int TRK_THREADS = 1024;
constexpr int TRK_MAX_D = 256;      // detections per frame
constexpr int TRK_MAX_M = 4096;     // memo entries visible to one match (tracklets + backdrops)

enum { H_NEXT_ID = 0, H_NTRK = 1, H_NBDF = 2, H_STATUS = 3, H_LAST_NV = 4, H_CALLS = 5, H_WORDS = 16 };

struct TrkState {
    int* hdr;                          // [H_WORDS]
    int *tid, *tlabel, *tlast, *tacc;  // [T]
    float *tbox, *tvel;                // [T][5]
    float* temb;                       // [T][E]
    int* bcnt;                         // [F]
    int* blabel;                       // [F][D]
    float* bbox;                       // [F][D][5]
    float* bemb;                       // [F][D][E]
};

struct TrkWs {
    float* semb;    // [D][E]   sorted, compacted detection embeddings
    float* score;   // [D][M]   similarity -> match scores
    float *cmax, *csum;                 // [M]
    int *stid, *stlabel, *stlast, *stacc;   // staging of the tracklet table for the expiry compaction
    float *stbox, *stvel, *stemb;
};

__host__ __device__ inline int trk_frames(const VknTrackerCfg& c) { return c.memo_backdrop_frames > 0 ? c.memo_backdrop_frames : 1; }
__host__ __device__ inline int trk_mmax(const VknTrackerCfg& c) { return c.max_tracklets + trk_frames(c) * c.max_dets; }

template <class T>
__host__ __device__ inline T* trk_take(char* base, size_t& off, size_t n) {
    off = (off + 255) & ~(size_t)255;
    T* p = reinterpret_cast<T*>(base + off);
    off += n * sizeof(T);
    return p;
}

__host__ __device__ inline size_t trk_carve(const VknTrackerCfg& c, char* base, TrkState* s) {
    const size_t T = c.max_tracklets, D = c.max_dets, E = c.embed_dim, F = trk_frames(c);
    size_t off = 0;
    s->hdr = trk_take<int>(base, off, H_WORDS);
    s->tid = trk_take<int>(base, off, T);
    s->tlabel = trk_take<int>(base, off, T);
    s->tlast = trk_take<int>(base, off, T);
    s->tacc = trk_take<int>(base, off, T);
    s->tbox = trk_take<float>(base, off, T * 5);
    s->tvel = trk_take<float>(base, off, T * 5);
    s->temb = trk_take<float>(base, off, T * E);
    s->bcnt = trk_take<int>(base, off, F);
    s->blabel = trk_take<int>(base, off, F * D);
    s->bbox = trk_take<float>(base, off, F * D * 5);
    s->bemb = trk_take<float>(base, off, F * D * E);
    return (off + 255) & ~(size_t)255;
}

__host__ __device__ inline size_t trk_carve_ws(const VknTrackerCfg& c, char* base, TrkWs* w) {
    const size_t T = c.max_tracklets, D = c.max_dets, E = c.embed_dim, M = trk_mmax(c);
    size_t off = 0;
    w->semb = trk_take<float>(base, off, D * E);
    w->score = trk_take<float>(base, off, D * M);
    w->cmax = trk_take<float>(base, off, M);
    w->csum = trk_take<float>(base, off, M);
    w->stid = trk_take<int>(base, off, T);
    w->stlabel = trk_take<int>(base, off, T);
    w->stlast = trk_take<int>(base, off, T);
    w->stacc = trk_take<int>(base, off, T);
    w->stbox = trk_take<float>(base, off, T * 5);
    w->stvel = trk_take<float>(base, off, T * 5);
    w->stemb = trk_take<float>(base, off, T * E);
    return (off + 255) & ~(size_t)255;
}

inline int trk_check(const VknTrackerCfg* c) {
    if (!c) return VKN_E_ARG;
    if (c->max_dets <= 0 || c->max_tracklets <= 0 || c->embed_dim <= 0 || c->memo_tracklet_frames < 0 || c->memo_backdrop_frames < 0 ||
        c->match_metric < 0 || c->match_metric > 2)
        return VKN_E_ARG;
    if (c->max_dets > TRK_MAX_D || trk_mmax(*c) > TRK_MAX_M || c->embed_dim > 1024) return VKN_E_SHAPE;
    return VKN_OK;
}

// mmdet `bbox_overlaps(mode='iou')` of two [x1, y1, x2, y2] boxes (eps 1e-6), every operation rounded on its own
__device__ __forceinline__ float trk_iou(const float* a, const float* b) {
    const float a1 = (a[2] - a[0]) * (a[3] - a[1]);
    const float a2 = (b[2] - b[0]) * (b[3] - b[1]);
    const float w = fmaxf(fminf(a[2], b[2]) - fmaxf(a[0], b[0]), 0.f);
    const float h = fmaxf(fminf(a[3], b[3]) - fmaxf(a[1], b[1]), 0.f);
    const float ov = w * h;
    const float un = fmaxf(a1 + a2 - ov, 1e-6f);
    return ov / un;
}

__global__ __launch_bounds__(TRK_THREADS) void k_qd_match(VknTrackerCfg cfg, char* state, const float* __restrict__ bboxes,
                                                            const long long* __restrict__ labels, const float* __restrict__ embeds,
                                                            int n, int frame_id, float* __restrict__ out_boxes,
                                                            long long* __restrict__ out_labels, long long* __restrict__ out_ids,
                                                            int* __restrict__ out_count, char* wsbuf) {
    __shared__ float s_score[TRK_MAX_D];
    __shared__ int s_order[TRK_MAX_D];         // sorted position -> input row
    __shared__ float s_box[TRK_MAX_D][5];      // sorted boxes
    __shared__ int s_flag[TRK_MAX_D];          // per-phase flags (valid / new / backdrop-kept)
    __shared__ float c_box[TRK_MAX_D][5];      // surviving boxes, score order
    __shared__ int c_src[TRK_MAX_D], c_label[TRK_MAX_D], c_id[TRK_MAX_D], c_slot[TRK_MAX_D], c_pos[TRK_MAX_D];
    __shared__ float r_max[TRK_MAX_D], r_sum[TRK_MAX_D];
    __shared__ int m_id[TRK_MAX_M], m_label[TRK_MAX_M];
    __shared__ volatile unsigned char m_taken[TRK_MAX_M];
    __shared__ int s_bdoff[66];
    __shared__ int s_nv, s_total, s_any;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int D = cfg.max_dets, E = cfg.embed_dim, T = cfg.max_tracklets, F = trk_frames(cfg);
    TrkState st;
    TrkWs ws;
    trk_carve(cfg, state, &st);
    trk_carve_ws(cfg, wsbuf, &ws);
    int ntrk = st.hdr[H_NTRK], nbdf = st.hdr[H_NBDF];
    const int next_id = st.hdr[H_NEXT_ID];
    int status = 0;

    // ---------------------------------------------------------------- A: stable rank sort by score, descending (:139-142)
    for (int i = tid; i < n; i += TRK_THREADS) s_score[i] = bboxes[i * 5 + 4];
    if (tid == 0) s_nv = 0;
    __syncthreads();
    if (tid < n) {
        const float si = s_score[tid];
        int r = 0;
        for (int j = 0; j < n; ++j) {
            const float sj = s_score[j];
            r += (sj > si) || (sj == si && j < tid);
        }
        s_order[r] = tid;
    }
    __syncthreads();
    for (int p = tid; p < n * 5; p += TRK_THREADS) s_box[p / 5][p % 5] = bboxes[s_order[p / 5] * 5 + p % 5];
    __syncthreads();
    // ---------------------------------------------------------------- B: duplicate suppression (:146-152): box i dies when ANY
    // better-scored box (kept or not) overlaps it by more than its class of threshold
    if (tid < n) {
        const float thr = s_box[tid][4] < cfg.obj_score_thr ? cfg.nms_backdrop_iou_thr : cfg.nms_class_iou_thr;
        int v = 1;
        for (int j = 0; j < tid; ++j)
            if (trk_iou(s_box[tid], s_box[j]) > thr) {
                v = 0;
                break;
            }
        s_flag[tid] = v;
    }
    __syncthreads();
    // ---------------------------------------------------------------- C: compaction + outputs (:153-159)
    if (tid < n) {
        int p = 0;
        for (int j = 0; j < tid; ++j) p += s_flag[j];
        const int v = s_flag[tid];
        if (v) {
            for (int e = 0; e < 5; ++e) {
                c_box[p][e] = s_box[tid][e];
                out_boxes[p * 5 + e] = s_box[tid][e];
            }
            const int src = s_order[tid];
            c_src[p] = src;
            const long long lb = labels[src];
            c_label[p] = (int)lb;
            out_labels[p] = lb;
            c_id[p] = -1;
            c_slot[p] = -1;
        }
        if (tid == n - 1) s_nv = p + v;
    }
    __syncthreads();
    const int nv = s_nv;
    for (int p = tid; p < nv * E; p += TRK_THREADS) ws.semb[p] = embeds[(size_t)c_src[p / E] * E + p % E];

    // ---------------------------------------------------------------- D: similarity against the memo (:162-185)
    // memo order = tracklets in creation order, then the backdrop frames newest first (`memo`, :105-135)
    int m = 0;
    const bool do_match = nv > 0 && ntrk > 0;      // `not self.empty`: backdrops alone do not make the memo non-empty
    if (do_match) {
        if (tid == 0) {
            int o = 0;
            for (int f = 0; f < nbdf; ++f) {
                s_bdoff[f] = o;
                o += st.bcnt[f];
            }
            s_bdoff[nbdf] = o;
        }
        __syncthreads();
        const int nbd = s_bdoff[nbdf];
        m = ntrk + nbd;
        auto memo_emb = [&](int j) -> const float* {
            if (j < ntrk) return st.temb + (size_t)j * E;
            int k = j - ntrk, f = 0;
            while (f + 1 < nbdf && k >= s_bdoff[f + 1]) ++f;
            return st.bemb + ((size_t)f * D + (k - s_bdoff[f])) * E;
        };
        for (int j = tid; j < m; j += TRK_THREADS) {
            if (j < ntrk) {
                m_id[j] = st.tid[j];
                m_label[j] = st.tlabel[j];
            } else {
                int k = j - ntrk, f = 0;
                while (f + 1 < nbdf && k >= s_bdoff[f + 1]) ++f;
                m_id[j] = -1;
                m_label[j] = st.blabel[f * D + (k - s_bdoff[f])];
            }
            m_taken[j] = 0;
        }
        __syncthreads();   // (also: semb is complete)
        const bool cosine = cfg.match_metric == 2;
        for (int p = tid; p < nv * m; p += TRK_THREADS) {
            const int i = p / m, j = p - i * m;
            const float* a = ws.semb + (size_t)i * E;
            const float* b = memo_emb(j);
            double acc = 0.0;
            if (!cosine) {
                for (int e = 0; e < E; ++e) acc = fma((double)a[e], (double)b[e], acc);   // (the fp64 product of two floats is exact)
            } else {   // F.normalize(x, p=2, dim=1) = x / max(||x||, 1e-12), then the product (:173-176)
                double na = 0.0, nb = 0.0;
                for (int e = 0; e < E; ++e) {
                    na = fma((double)a[e], (double)a[e], na);
                    nb = fma((double)b[e], (double)b[e], nb);
                }
                const float da = fmaxf((float)sqrt(na), 1e-12f), db = fmaxf((float)sqrt(nb), 1e-12f);
                for (int e = 0; e < E; ++e) acc = fma((double)(a[e] / da), (double)(b[e] / db), acc);
            }
            ws.score[p] = (float)acc;
        }
        __syncthreads();
        if (!cosine) {
            // d2t = softmax over the memo axis, t2d = softmax over the detection axis (bisoftmax only)      (:164-171)
            if (tid < nv) {
                const float* row = ws.score + (size_t)tid * m;
                float mx = -INFINITY;
                for (int j = 0; j < m; ++j) mx = fmaxf(mx, row[j]);
                float sm = 0.f;
                for (int j = 0; j < m; ++j) sm += expf(row[j] - mx);
                r_max[tid] = mx;
                r_sum[tid] = sm;
            }
            if (cfg.match_metric == 0) {
                for (int j = tid; j < m; j += TRK_THREADS) {
                    float mx = -INFINITY;
                    for (int i = 0; i < nv; ++i) mx = fmaxf(mx, ws.score[(size_t)i * m + j]);
                    float sm = 0.f;
                    for (int i = 0; i < nv; ++i) sm += expf(ws.score[(size_t)i * m + j] - mx);
                    ws.cmax[j] = mx;
                    ws.csum[j] = sm;
                }
            }
            __syncthreads();
        }
        for (int p = tid; p < nv * m; p += TRK_THREADS) {
            const int i = p / m, j = p - i * m;
            float s = ws.score[p];
            if (!cosine) {
                const float d2t = expf(s - r_max[i]) / r_sum[i];
                if (cfg.match_metric == 0) {
                    const float t2d = expf(s - ws.cmax[j]) / ws.csum[j];
                    s = (d2t + t2d) / 2.f;
                } else {
                    s = d2t;
                }
            }
            if (cfg.with_cats) s *= (c_label[i] == m_label[j]) ? 1.f : 0.f;   // (:178-180)
            ws.score[p] = s;
        }
        __syncthreads();
        // ------------------------------------------------------------ E: greedy assignment, one wave (:182-196)
        if (wave == 0) {
            for (int i = 0; i < nv; ++i) {
                const float* row = ws.score + (size_t)i * m;
                float best = -INFINITY;
                int bj = 0x7fffffff;
                for (int j = lane; j < m; j += 64) {
                    const float v = m_taken[j] ? 0.f : row[j];   // a taken track's column reads 0 for every other detection
                    if (v > best) {
                        best = v;
                        bj = j;
                    }
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    const float ov = __shfl_xor(best, o, 64);
                    const int oj = __shfl_xor(bj, o, 64);
                    if (ov > best || (ov == best && oj < bj)) {
                        best = ov;
                        bj = oj;
                    }
                }
                if (lane == 0 && best > cfg.match_score_thr) {
                    const int id = m_id[bj];
                    if (id > -1) {
                        if (c_box[i][4] > cfg.obj_score_thr) {
                            c_id[i] = id;
                            c_slot[i] = bj;
                            m_taken[bj] = 1;
                        } else if (best > cfg.nms_conf_thr) {
                            c_id[i] = -2;
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        __syncthreads();
    }
    // ---------------------------------------------------------------- F: births (:197-203)
    if (tid < nv) s_flag[tid] = (c_id[tid] == -1 && c_box[tid][4] > cfg.init_score_thr) ? 1 : 0;
    __syncthreads();
    int my_new = -1;
    if (tid < nv) {
        int p = 0;
        for (int j = 0; j < tid; ++j) p += s_flag[j];
        if (s_flag[tid]) {
            my_new = p;
            c_id[tid] = next_id + p;
        }
        if (tid == nv - 1) s_total = p + s_flag[tid];
    }
    if (nv == 0 && tid == 0) s_total = 0;
    __syncthreads();
    const int total_new = s_total;
    if (tid < nv) out_ids[tid] = c_id[tid];

    // ---------------------------------------------------------------- G: memo update (`update_memo`, :47-103)
    // matched tracks: box, momentum embedding, label, running mean of the velocity
    for (int p = tid; p < nv * 5; p += TRK_THREADS) {
        const int i = p / 5, e = p - i * 5, t = c_slot[i];
        if (t >= 0) {
            const float dt = (float)(frame_id - st.tlast[t]);
            const float acc = (float)st.tacc[t];
            const float vel = (c_box[i][e] - st.tbox[t * 5 + e]) / dt;
            st.tvel[t * 5 + e] = (st.tvel[t * 5 + e] * acc + vel) / (acc + 1.f);
            st.tbox[t * 5 + e] = c_box[i][e];
        }
    }
    for (int p = tid; p < nv * E; p += TRK_THREADS) {
        const int i = p / E, e = p - i * E, t = c_slot[i];
        if (t >= 0) st.temb[(size_t)t * E + e] = cfg.memo_keep * st.temb[(size_t)t * E + e] + cfg.memo_momentum * ws.semb[p];
    }
    __syncthreads();   // (tlast / tacc are read above, written below)
    if (tid < nv) {
        const int t = c_slot[tid];
        if (t >= 0) {
            st.tlast[t] = frame_id;
            st.tlabel[t] = c_label[tid];
            st.tacc[t] += 1;
        }
        if (my_new >= 0) {
            const int tn = ntrk + my_new;
            if (tn < T) {
                st.tid[tn] = c_id[tid];
                st.tlabel[tn] = c_label[tid];
                st.tlast[tn] = frame_id;
                st.tacc[tn] = 0;
                for (int e = 0; e < 5; ++e) {
                    st.tbox[tn * 5 + e] = c_box[tid][e];
                    st.tvel[tn * 5 + e] = 0.f;
                }
                c_pos[tid] = tn;
            } else {
                c_pos[tid] = -1;
                status |= 1;   // tracklet table full: the birth is dropped (the id is still consumed)
            }
        } else {
            c_pos[tid] = -1;
        }
    }
    __syncthreads();
    for (int p = tid; p < nv * E; p += TRK_THREADS) {
        const int i = p / E, tn = c_pos[i];
        if (tn >= 0) st.temb[(size_t)tn * E + (p - i * E)] = ws.semb[p];
    }
    ntrk = min(ntrk + total_new, T);
    // backdrops: still-unassigned detections that no better-scored detection overlaps (:81-93), newest frame first
    if (tid < nv) {
        int keep = c_id[tid] == -1;
        if (keep)
            for (int j = 0; j < tid; ++j)
                if (trk_iou(c_box[tid], c_box[j]) > cfg.nms_backdrop_iou_thr) {
                    keep = 0;
                    break;
                }
        s_flag[tid] = keep;
    }
    __syncthreads();
    if (cfg.memo_backdrop_frames > 0) {
        const int keepf = min(nbdf, F - 1);   // frames that survive the insert (`backdrops.pop()` beyond memo_backdrop_frames)
        for (int f = keepf; f >= 1; --f) {    // frame f - 1 -> f, oldest first
            const int cnt = st.bcnt[f - 1];
            for (int p = tid; p < cnt * E; p += TRK_THREADS) st.bemb[(size_t)f * D * E + p] = st.bemb[(size_t)(f - 1) * D * E + p];
            for (int p = tid; p < cnt * 5; p += TRK_THREADS) st.bbox[(size_t)f * D * 5 + p] = st.bbox[(size_t)(f - 1) * D * 5 + p];
            for (int p = tid; p < cnt; p += TRK_THREADS) st.blabel[f * D + p] = st.blabel[(f - 1) * D + p];
            __syncthreads();
            if (tid == 0) st.bcnt[f] = cnt;
            __syncthreads();
        }
        if (tid < nv) {
            int p = 0;
            for (int j = 0; j < tid; ++j) p += s_flag[j];
            c_pos[tid] = s_flag[tid] ? p : -1;
            if (s_flag[tid]) {
                st.blabel[p] = c_label[tid];
                for (int e = 0; e < 5; ++e) st.bbox[p * 5 + e] = c_box[tid][e];
            }
            if (tid == nv - 1) st.bcnt[0] = p + s_flag[tid];
        }
        if (nv == 0 && tid == 0) st.bcnt[0] = 0;
        __syncthreads();
        for (int p = tid; p < nv * E; p += TRK_THREADS) {
            const int i = p / E, b = c_pos[i];
            if (b >= 0) st.bemb[(size_t)b * E + (p - i * E)] = ws.semb[p];
        }
        nbdf = min(nbdf + 1, F);
    }
    // expiry: tracks not seen for memo_tracklet_frames frames leave the table; the others keep their order (:95-100)
    if (tid == 0) s_any = 0;
    __syncthreads();
    for (int t = tid; t < ntrk; t += TRK_THREADS)
        if (frame_id - st.tlast[t] >= cfg.memo_tracklet_frames) s_any = 1;
    __syncthreads();
    if (s_any) {
        for (int t = tid; t < ntrk; t += TRK_THREADS) {
            ws.stid[t] = st.tid[t];
            ws.stlabel[t] = st.tlabel[t];
            ws.stlast[t] = st.tlast[t];
            ws.stacc[t] = st.tacc[t];
        }
        for (int p = tid; p < ntrk * 5; p += TRK_THREADS) {
            ws.stbox[p] = st.tbox[p];
            ws.stvel[p] = st.tvel[p];
        }
        for (int p = tid; p < ntrk * E; p += TRK_THREADS) ws.stemb[p] = st.temb[p];
        __syncthreads();
        // destination of row t = number of surviving rows before it; m_id doubles as the position table (the match is over)
        int* pos = m_id;
        if (tid == 0) {
            int o = 0;
            for (int t = 0; t < ntrk; ++t) {
                const bool keep = frame_id - ws.stlast[t] < cfg.memo_tracklet_frames;
                pos[t] = keep ? o : -1;
                o += keep;
            }
            s_total = o;
        }
        __syncthreads();
        for (int t = tid; t < ntrk; t += TRK_THREADS) {
            const int d = pos[t];
            if (d >= 0) {
                st.tid[d] = ws.stid[t];
                st.tlabel[d] = ws.stlabel[t];
                st.tlast[d] = ws.stlast[t];
                st.tacc[d] = ws.stacc[t];
            }
        }
        for (int p = tid; p < ntrk * 5; p += TRK_THREADS) {
            const int d = pos[p / 5];
            if (d >= 0) {
                st.tbox[d * 5 + p % 5] = ws.stbox[p];
                st.tvel[d * 5 + p % 5] = ws.stvel[p];
            }
        }
        for (int p = tid; p < ntrk * E; p += TRK_THREADS) {
            const int d = pos[p / E];
            if (d >= 0) st.temb[(size_t)d * E + p % E] = ws.stemb[p];
        }
        ntrk = s_total;
    }
    // any thread may have seen the overflow
    // out_count[1] reports THIS call's status; the header keeps the sticky union for introspection (`tracker.status`).  Reporting
    // the sticky word made every later match() raise until reset().
    __shared__ int s_call_status;
    if (tid == 0) s_call_status = 0;
    __syncthreads();
    if (status) {
        atomicOr(&s_call_status, status);
        atomicOr(&st.hdr[H_STATUS], status);
    }
    __syncthreads();
    if (tid == 0) {
        st.hdr[H_NEXT_ID] = next_id + total_new;
        st.hdr[H_NTRK] = ntrk;
        st.hdr[H_NBDF] = nbdf;
        st.hdr[H_LAST_NV] = nv;
        st.hdr[H_CALLS] += 1;
        out_count[0] = nv;
        out_count[1] = s_call_status;
    }
}

}  // namespace

extern "C" {

size_t vkn_sizeof_tracker_cfg(void) { return sizeof(VknTrackerCfg); }

size_t vkn_qd_tracker_state_bytes(const VknTrackerCfg* cfg) {
    if (trk_check(cfg) != VKN_OK) return 0;
    TrkState s;
    return trk_carve(*cfg, nullptr, &s);
}

size_t vkn_qd_tracker_workspace_bytes(const VknTrackerCfg* cfg) {
    if (trk_check(cfg) != VKN_OK) return 0;
    TrkWs w;
    return trk_carve_ws(*cfg, nullptr, &w);
}

int vkn_qd_tracker_state_layout(const VknTrackerCfg* cfg, size_t* offsets) {
    const int rc = trk_check(cfg);
    if (rc != VKN_OK) return rc;
    if (!offsets) return VKN_E_ARG;
    TrkState s;
    trk_carve(*cfg, nullptr, &s);
    const void* p[12] = {s.hdr, s.tid, s.tlabel, s.tlast, s.tacc, s.tbox, s.tvel, s.temb, s.bcnt, s.blabel, s.bbox, s.bemb};
    for (int i = 0; i < 12; ++i) offsets[i] = reinterpret_cast<size_t>(p[i]);
    return VKN_OK;
}

int vkn_qd_tracker_reset(const VknTrackerCfg* cfg, void* state, size_t state_bytes, void* stream) {
    const int rc = trk_check(cfg);
    if (rc != VKN_OK) return rc;
    if (!state) return VKN_E_ARG;
    if (state_bytes < vkn_qd_tracker_state_bytes(cfg)) return VKN_E_WORKSPACE;
    // header (ids, counts, status) and the backdrop counters; the tables behind them are dead data once the counts are zero
    TrkState s;
    trk_carve(*cfg, static_cast<char*>(state), &s);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (hipMemsetAsync(s.hdr, 0, H_WORDS * sizeof(int), st) != hipSuccess) return VKN_E_LAUNCH;
    if (hipMemsetAsync(s.bcnt, 0, trk_frames(*cfg) * sizeof(int), st) != hipSuccess) return VKN_E_LAUNCH;
    return VKN_OK;
}

int vkn_qd_tracker_match_f32(const VknTrackerCfg* cfg, void* state, size_t state_bytes, const float* bboxes, const int64_t* labels,
                             const float* embeds, int n, int frame_id, float* out_bboxes, int64_t* out_labels, int64_t* out_ids,
                             int* out_count, void* ws, size_t ws_bytes, void* stream) {
    const int rc = trk_check(cfg);
    if (rc != VKN_OK) return rc;
    if (!state || !out_count || n < 0) return VKN_E_ARG;
    if (n > 0 && (!bboxes || !labels || !embeds || !out_bboxes || !out_labels || !out_ids)) return VKN_E_ARG;
    if (n > cfg->max_dets) return VKN_E_SHAPE;
    if (state_bytes < vkn_qd_tracker_state_bytes(cfg)) return VKN_E_WORKSPACE;
    if (!ws || ws_bytes < vkn_qd_tracker_workspace_bytes(cfg)) return VKN_E_WORKSPACE;
    if ((reinterpret_cast<uintptr_t>(state) & 255) || (reinterpret_cast<uintptr_t>(ws) & 255)) return VKN_E_ALIGN;
    hipLaunchKernelGGL(k_qd_match, dim3(1), dim3(TRK_THREADS), 0, static_cast<hipStream_t>(stream), *cfg, static_cast<char*>(state),
                       bboxes, reinterpret_cast<const long long*>(labels), embeds, n, frame_id, out_bboxes,
                       reinterpret_cast<long long*>(out_labels), reinterpret_cast<long long*>(out_ids), out_count,
                       static_cast<char*>(ws));
    VKN_CHECK_LAUNCH();
    return VKN_OK;
}

}  // extern "C"
