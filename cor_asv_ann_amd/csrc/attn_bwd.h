// Attention backward for ONE sample of one decoder time step (oracle/train.py; no gradient through the window mask nor through
// the previous alignment, attention.py:567), worked by a group of `nthr` threads (whole waves) of a workgroup: all 256 of the
// per-step kernel's (train_kernels.hip), or half of a persistent workgroup that handles two samples side by side
// (train_persist_topb.hip).  Every group of the workgroup must call it (its barriers are the workgroup's); `active` = false
// makes a group without a sample keep step only.
// Everything that is read is independent of what is written, so the loads batch; the read-modify-writes of d_enc / du go out as
// fire-and-forget float atomics (distinct addresses per sample: no contention).  dva / dbv are kept as per-sample partial sums
// across the steps and reduced once after the loop.
// HANDOFF: dL/dctx was accumulated by other workgroups of the same launch (read past the L1), dwq is read by them (write-through).
// DEFER (persistent recurrence, round 4): d_enc / du are NOT touched here -- the step leaves its dL/dscore row (11 floats) in
// p.ds_out, and two whole-sequence kernels behind the recurrence (attention_deferred_*_kernel, train_kernels.hip) add up
// d_enc[s] = sum_t a_t[s] dctx_t and du[s] = sum_t dscore_t[s] v_a (1 - tanh^2(wq_t + u_s)) per sample in LDS, from what the
// passes keep per step anyway: 2 x 5 632 float atomics per sample and step (23 MB per step at configs[3], drained by the memory
// side at ~1.3 TB/s in the middle of the recurrence's critical path) become one pass over ~0.3 GB at the end.
#pragma once
#include "common.h"
#include "row_kernels.h"
#include "train_kernels.h"

namespace casv {

template <bool HANDOFF, int DEFER = 0>       // DEFER bits: 1 = d_enc, 2 = du summed behind the recurrence
__device__ __forceinline__ void attention_bwd_sample(const AttnBwdArgs& p, const int b, const bool active, const int tid, const int nthr,
                                                     float* s_dx, float* s_da, float* s_ds, float* s_av) {
    const int lane = tid & 63, wave = tid >> 6, nwaves = nthr >> 6;
    const int W = p.W, C = p.C, T = p.T;
    int s_lo = 0, cnt = 0;
    if (active) {
        const int wv = p.win[b];
        s_lo = wv & 0xffff; cnt = wv >> 16;
        const float* dx = p.dxh + (long long)b * p.ld_dxh + p.ctx_off;
        const float* mc = p.mcell ? p.mcell + (long long)b * p.ld_mcell + p.mc_off : nullptr;
        for (int c = tid; c < C; c += nthr) {
            const float d = HANDOFF ? __hip_atomic_load(dx + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : dx[c];
            s_dx[c] = d * (mc ? mc[c] : 1.0f);
        }
        if (tid < 16) s_av[tid] = tid < cnt ? p.a[(long long)b * T + s_lo + tid] : 0.0f;
    }
    __syncthreads();
    // da_s = dctx . enc_s
    for (int i = wave; i < cnt; i += nwaves) {
        const float* es = p.enc + (long long)b * p.enc_line + (long long)(s_lo + i) * p.enc_time;
        float part = 0.f;
        for (int c = lane; c < C; c += 64) part += s_dx[c] * es[c];
        part = wave_sum(part);
        if (lane == 0) s_da[i] = part;
    }
    __syncthreads();
    if (active && tid < 16) {
        float dot = 0.f;
        for (int i = 0; i < cnt; ++i) dot += s_av[i] * s_da[i];
        s_ds[tid] = tid < cnt ? s_av[tid] * (s_da[tid] - dot) : 0.0f;          // dL/dscore
    }
    __syncthreads();
    if (!active) return;
    if (DEFER && tid < 16) p.ds_out[(long long)b * 16 + tid] = s_ds[tid];
    if (!(DEFER & 1)) {
        // d enc_out[s] += a_s * dctx
        float* de = p.d_enc + (long long)b * p.enc_line + (long long)s_lo * p.enc_time;
        for (int i = 0; i < cnt; ++i) {
            const float av = s_av[i];
            for (int c = tid; c < C; c += nthr) atomicAdd(de + (long long)i * p.enc_time + c, av * s_dx[c]);
        }
    }
    // energies: th = tanh(wq + u_s); dva += dscore*th ; dpre = dscore*va*(1-th^2) -> du_s, dwq
    for (int j = tid; j < W; j += nthr) {
        const float q = p.wq[(long long)b * W + j], v = p.va[j];
        const long long off0 = (long long)b * p.u_line + (long long)s_lo * p.u_time + j;
        float uu[11];
#pragma unroll
        for (int i = 0; i < 11; ++i) uu[i] = p.u[off0 + (long long)(i < cnt ? i : 0) * p.u_time];
        float dwq = 0.f, dva = 0.f;
#pragma unroll
        for (int i = 0; i < 11; ++i) {
            const float th = fast_tanh(q + uu[i]);
            const float ds = s_ds[i];
            dva += ds * th;
            const float dpre = ds * v * (1.0f - th * th);
            if (!(DEFER & 2) && i < cnt) atomicAdd(p.du + off0 + (long long)i * p.u_time, dpre);
            dwq += dpre;
        }
        if (HANDOFF) store_sc1(p.dwq + (long long)b * W + j, dwq);
        else p.dwq[(long long)b * W + j] = dwq;
        p.dva_part[(long long)b * W + j] += dva;
    }
    if (tid == 0) {
        float dbv = 0.f;
        for (int i = 0; i < cnt; ++i) dbv += s_ds[i];
        p.dbv_part[b] += dbv;
    }
}

}  // namespace casv
