// sampler.hip — classifier-free guidance + top-k + top-p + multinomial(1) in one launch, one 256-thread workgroup per
// (image, token) row of V logits.  Replaces var.py:172-175 + helpers.py:6-19 (≈6 ATen launches, a full sort and a
// cumsum over every row) and consumes the same Exp(1) noise tensor torch.multinomial would draw.
//
// Integer-exact by construction: top-k is a radix select on the monotone integer image of the floats, the top-p cut
// is the one the walk over the stable ascending order (bitonic sort of (key,index) pairs, only over the entries that survived
// top-k) with the fp64 running sum ATen's cumsum uses would make — decided by a parallel prefix sum wherever that is provably the
// same decision, by the walk itself otherwise — and the two softmax denominators use the canonical W256 sum.
#include "common.h"

__device__ __forceinline__ unsigned long long shfl_u64(unsigned long long v, int src) {
    return ((unsigned long long)__shfl((unsigned)(v >> 32), src, 64) << 32) | (unsigned)__shfl((unsigned)v, src, 64);
}

// block-wide inclusive prefix sums over the 256 threads (thread order); `sw` holds 4 slots; contains one barrier
__device__ __forceinline__ int vs_scan256_i(int v, int* sw) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(v, off, 64); if (lane >= off) v += o; }
    if (lane == 63) sw[wave] = v;
    __syncthreads();
    int add = 0;
    for (int w = 0; w < wave; ++w) add += sw[w];
    return v + add;
}
__device__ __forceinline__ double vs_scan256_d(double v, double* sw) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const double o = __shfl_up(v, off, 64); if (lane >= off) v += o; }
    if (lane == 63) sw[wave] = v;
    __syncthreads();
    double add = 0.0;
    for (int w = 0; w < wave; ++w) add += sw[w];
    return v + add;
}

__device__ __forceinline__ unsigned long long vs_scan256_u64(unsigned long long v, unsigned long long* sw) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned lo = __shfl_up((unsigned)v, off, 64), hi = __shfl_up((unsigned)(v >> 32), off, 64);
        if (lane >= off) v += ((unsigned long long)hi << 32) | lo;
    }
    if (lane == 63) sw[wave] = v;
    __syncthreads();
    unsigned long long add = 0ull;
    for (int w = 0; w < wave; ++w) add += sw[w];
    return v + add;
}

// NV = V / 256 when the caller's V is the common 4096 (the exponentials of a thread's NV elements then stay in registers between the
// phases: one evaluation instead of three), 0 = any V (evaluated where needed)
template <int NV>
__global__ void __launch_bounds__(256) k_cfg_sample(const float* __restrict__ logits, const float* __restrict__ noise, int64_t* __restrict__ idx_out,
                                                    float* __restrict__ masked_out, int64_t rows, int V, float ca, float cb,
                                                    int top_k, int use_top_p, float thr, int cap, int force_walk) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm_raw[];
    float* xs = reinterpret_cast<float*>(sm_raw);                                   // [V] working logits
    unsigned long long* srt = reinterpret_cast<unsigned long long*>(sm_raw + sizeof(float) * V);   // [cap] (key<<32 | idx), cap = pow2 >= top_k (or V)
    __shared__ __attribute__((aligned(8))) int s_hist[2][256];      // top-k: two count tables; top-p (3a): one table of 256 doubles (the probability mass per byte value)
    __shared__ int s_ci[8];
    __shared__ double s_cd[4];
    __shared__ float red[4];
    __shared__ unsigned s_cnt;
    __shared__ int s_sel[2];
    __shared__ int s_flag;
    __shared__ float s_bv[4]; __shared__ int s_bi[4]; __shared__ int s_bn[4];
    __shared__ double s_c0; __shared__ int s_pick; __shared__ int s_fast; __shared__ int s_ntie; __shared__ unsigned s_tie[64];     // top-p by selection (3a)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t row = blockIdx.x;
    const float* lc = logits + row * V;
    const float* lu = logits + (rows + row) * V;

    // (1) CFG combine: (1+t)*cond - t*uncond, three roundings as in the reference's tensor expression
    // (16 bytes per lane; which thread computes an element is irrelevant here — the sums below read xs in their own canonical assignment)
    for (int i4 = tid; i4 < (V >> 2); i4 += 256) {
        const f32x4 c4 = *(const f32x4*)(lc + 4 * i4), u4 = *(const f32x4*)(lu + 4 * i4);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float a = ca * c4[e]; const float b = cb * u4[e]; o[e] = a - b; }
        *(f32x4*)(xs + 4 * i4) = o;
    }
    // the row's Exp(1) noise is needed last: requested now (NV > 0: into registers), it arrives behind everything else
    float qv[NV > 0 ? NV : 1];
    if constexpr (NV > 0) {
#pragma unroll
        for (int k = 0; k < NV; ++k) qv[k] = noise[row * V + tid + 256 * k];
    }
    s_hist[0][tid] = 0;
    __syncthreads();

    // (2) top-k: T = key of the k-th largest, by an 8-bit radix select on the monotone integer image of the floats (four passes: count
    // the candidates per value of the current byte, walk the 256 counts from the top).  Round 2 built T bit by bit: 32 block reductions.
    unsigned T = 0u;                                   // (0: no top-k, every key is >= it)
    if (top_k > 0) {
        unsigned prefix = 0u, mask = 0u;
        int need = top_k;                              // T is the need-th largest of the keys with (key & mask) == prefix
#pragma unroll 1
        for (int pass = 0; pass < 4; ++pass) {
            const int shift = 24 - 8 * pass;
            int* hist = s_hist[pass & 1];
            s_hist[(pass + 1) & 1][tid] = 0;           // the other table: next pass (nobody reads it before two barriers from here)
            for (int i = tid; i < V; i += 256) {
                const unsigned key = vm_float_key(xs[i]);
                if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1);
            }
            __syncthreads();
            const int mine = hist[255 - tid];          // thread t owns byte value 255 - t: the inclusive scan is the count of candidates >= that value
            const int incl = vs_scan256_i(mine, s_ci);
            if (incl >= need && incl - mine < need) { s_sel[0] = 255 - tid; s_sel[1] = incl - mine; }
            __syncthreads();
            prefix |= (unsigned)s_sel[0] << shift; mask |= 255u << shift; need -= s_sel[1];
        }
        T = prefix;
        for (int i = tid; i < V; i += 256) if (vm_float_key(xs[i]) < T) xs[i] = -INFINITY;
        __syncthreads();
    }

    // row max (exact in any order)
    float m = -INFINITY;
    for (int i = tid; i < V; i += 256) m = fmaxf(m, xs[i]);
    m = vh_block_max256(m, red);
    float ev[NV > 0 ? NV : 1];                                      // (NV > 0) exp(x - m) of this thread's elements

    // (3) top-p
    if (use_top_p) {
        float part = 0.f;
        int fin = 0;
        if constexpr (NV > 0) {
#pragma unroll
            for (int k = 0; k < NV; ++k) { const float x = xs[tid + 256 * k]; ev[k] = vm_exp(x - m); part = part + ev[k]; fin += (x > -INFINITY) ? 1 : 0; }
        } else
        for (int i = tid; i < V; i += 256) { part = part + vm_exp(xs[i] - m); fin += (xs[i] > -INFINITY) ? 1 : 0; }
        const float S = vh_block_sum256(part, red);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) fin += __shfl_xor(fin, off, 64);
        if (lane == 0) s_ci[4 + wave] = fin;
        if (tid == 0) { s_cnt = 0u; s_flag = 0; }
        __syncthreads();
        const int total = s_ci[4] + s_ci[5] + s_ci[6] + s_ci[7];      // survivors of top-k (ties with the k-th value included)
        // The sort buffer holds `cap` entries (a power of two >= top_k, so that more workgroups fit a CU).  Only a crowd of exact
        // ties with the k-th value can exceed it; those ties are the SMALLEST survivors and sort among themselves by index, so
        // they need no sorting: in that case only the entries strictly above the k-th value (< top_k of them) are sorted and the
        // tie group is walked in index order first.
        const bool split = total > cap;
        // compact (order irrelevant: they get sorted): one counter update per wave and step (ballot), not one per entry
        for (int i0 = 0; i0 < V; i0 += 256) {
            const int i = i0 + tid;
            const float v = xs[i];
            const bool take = v > -INFINITY && (!split || vm_float_key(v) > T);
            const unsigned long long bal = __ballot(take);
            unsigned base = 0u;
            if (lane == 0 && bal) base = atomicAdd(&s_cnt, (unsigned)__popcll(bal));
            base = __shfl(base, 0, 64);
            if (take) srt[base + __popcll(bal & ((1ull << lane) - 1ull))] = ((unsigned long long)vm_float_key(v) << 32) | (unsigned)i;
        }
        __syncthreads();
        const int cnt = (int)s_cnt;
        // (3a) The cut WITHOUT sorting.  The removed set is a prefix of the ascending (key, index) order: every entry whose running sum
        // c (fp64, ascending) satisfies (float)c <= thr, i.e. c below `mid`, the midpoint of thr and the next float.  That prefix is found
        // by selection, the way top-k found its threshold: four passes over the key bytes, each summing the probability mass per byte value
        // and walking the 256 sums upwards to the value where the running mass crosses mid.  Result: the key K* of the entry at which the
        // running sum crosses, and C0 = the mass of everything below K*.  Entries below K* go; the (normally one) entries equal to K* are
        // taken in index order.  The sums are formed in another order than the sequential walk's, so they differ from its running sums by
        // at most n * 2^-53 (relative): the decision is the walk's whenever the two running sums next to the cut — the last removed
        // entry's and the first kept one's — are farther than 1e-9 (relative) from mid, which is checked; otherwise, and for tie crowds,
        // the sort + walk below decides as before (varhip_sampler_force_walk(1) forces it: the tests compare both against the oracle).
        bool cut_done = false;
        if (!split && thr > 1e-30f && cnt >= 2 && !force_walk) {
            const double mid = 0.5 * ((double)thr + (double)__uint_as_float(__float_as_uint(thr) + 1u));
            // how close to mid a running sum may come before the walk has to decide: 1e-9 relative covers the reordering of the additions
            // (n * 2^-53 relative), cnt * 2^-61 absolute covers the truncation of cnt probabilities to the 2^-62 grid — the larger of the two
            // (for thr = 1 - top_p below ~4e-6, i.e. top_p within a few 1e-6 of one, the absolute term is the wider)
            const double band = fmax(1e-9 * mid, (double)cnt * 4.336808689942018e-19);
            // The mass per byte value is summed in 2^-62 fixed point: integer LDS atomics do not mind the contention that the survivors'
            // clustered leading bytes put on a few addresses (fp64 atomics on one address serialise badly), the sums are exact in that grid and
            // the same in every run; rounding a probability to the grid moves a sum by <= n * 2^-62 absolute — inside `band` above.
            unsigned long long* const s_hs = reinterpret_cast<unsigned long long*>(&s_hist[0][0]);
            unsigned long long* const s_cu = reinterpret_cast<unsigned long long*>(s_cd);
            const double FX = 4611686018427387904.0, IFX = 1.0 / 4611686018427387904.0;       // 2^62
            // the probability of an entry from its key (vm_float_key is invertible: -0 was folded into +0, which has the same exponential)
            auto prob_of = [&](unsigned key) { const unsigned u = (key & 0x80000000u) ? (key & 0x7FFFFFFFu) : ~key; return vm_exp(__uint_as_float(u) - m) / S; };
            unsigned prefix = 0u, mask = 0u;
            unsigned long long C0u = 0ull;
            double C0 = 0.0;
            bool ok = true;
#pragma unroll 1
            for (int pass = 0; pass < 4 && ok; ++pass) {
                const int shift = 24 - 8 * pass;
                s_hs[tid] = 0ull;
                if (tid == 0) s_pick = -1;
                __syncthreads();
                for (int e = tid; e < cnt; e += 256) {
                    const unsigned key = (unsigned)(srt[e] >> 32);
                    if ((key & mask) == prefix) atomicAdd(&s_hs[(key >> shift) & 255u], (unsigned long long)((double)prob_of(key) * FX));
                }
                __syncthreads();
                const unsigned long long mine = s_hs[tid];
                const unsigned long long incl = vs_scan256_u64(mine, s_cu);   // thread t owns byte value t: mass of the candidates with a byte <= t
                if (mine > 0ull && (double)(C0u + incl) * IFX > mid && !((double)(C0u + (incl - mine)) * IFX > mid)) { s_pick = tid; s_c0 = (double)(C0u + (incl - mine)) * IFX; s_hs[0] = C0u + (incl - mine); }
                __syncthreads();
                if (s_pick < 0) ok = false;                              // (the whole mass stays below mid: degenerate, the walk decides)
                else { prefix |= (unsigned)s_pick << shift; mask |= 255u << shift; C0 = s_c0; C0u = s_hs[0]; }
                __syncthreads();
            }
            if (tid == 0) { s_ntie = 0; s_fast = ok ? 1 : 0; }
            __syncthreads();
            if (ok) {
                // entries equal to K* (ties share one probability): listed, ranked by index, each one's running sum formed by repeated addition
                for (int e = tid; e < cnt; e += 256)
                    if ((unsigned)(srt[e] >> 32) == prefix) { const int q = atomicAdd(&s_ntie, 1); if (q < 64) s_tie[q] = (unsigned)srt[e]; }
                __syncthreads();
                const int ntie = s_ntie;
                if (ntie > 64 || fabs(C0 - mid) <= band) { if (tid == 0) s_fast = 0; }
                __syncthreads();
                unsigned rm = 0u;                                        // bit k: this thread's k-th entry is removed
                if (s_fast) {
                    const double pT = (double)prob_of(prefix);
                    int k = 0;
                    for (int e = tid; e < cnt; e += 256, ++k) {
                        const unsigned key = (unsigned)(srt[e] >> 32), idx = (unsigned)srt[e];
                        if (key < prefix) rm |= 1u << k;
                        else if (key == prefix) {
                            int rank = 0;
                            for (int q = 0; q < ntie; ++q) rank += (s_tie[q] < idx) ? 1 : 0;
                            double c = C0;
                            for (int q = 0; q <= rank; ++q) c += pT;
                            if (fabs(c - mid) <= band) atomicAnd(&s_fast, 0);
                            else if (c < mid) rm |= 1u << k;
                        }
                    }
                }
                __syncthreads();
                if (s_fast) {
                    int k = 0;
                    for (int e = tid; e < cnt; e += 256, ++k) if ((rm >> k) & 1u) xs[(unsigned)srt[e]] = -INFINITY;
                    cut_done = true;
                }
            }
            __syncthreads();
        }
        if (!cut_done) {
        int n2 = 2; while (n2 < cnt) n2 <<= 1;
        for (int i = cnt + tid; i < n2; i += 256) srt[i] = ~0ull;       // pad to a power of two with +max keys
        __syncthreads();
        // bitonic sort, ascending (key, index).  A wave owns a quarter of the entries: every stage whose partner distance j stays inside a
        // quarter needs no workgroup barrier (a wave's LDS operations complete in order): 52 of the 55 stages at 1024 entries.
        const int bs = n2 >= 256 ? (n2 >> 2) : n2;                      // entries per wave block (small sorts: wave 0 alone)
        const bool active = n2 >= 256 || wave == 0;
        for (int k = 2; k <= n2; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                if (j >= bs) {                                           // partners in different waves' blocks
                    __syncthreads();
                    for (int i = tid; i < n2; i += 256) {
                        const int ixj = i ^ j;
                        if (ixj > i) {
                            const unsigned long long a = srt[i], b = srt[ixj];
                            if ((a > b) == ((i & k) == 0)) { srt[i] = b; srt[ixj] = a; }
                        }
                    }
                    __syncthreads();
                } else if (active) {
                    for (int i = wave * bs + lane; i < (wave + 1) * bs && i < n2; i += 64) {
                        const int ixj = i ^ j;
                        if (ixj > i) {
                            const unsigned long long a = srt[i], b = srt[ixj];
                            if ((a > b) == ((i & k) == 0)) { srt[i] = b; srt[ixj] = a; }
                        }
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's exchanges have landed before its next stage reads them
                }
            }
        }
        __syncthreads();
        // every sorted survivor's probability replaces the (now useless) key half of its sort entry ...
        for (int sidx = tid; sidx < cnt; sidx += 256) {
            const unsigned i = (unsigned)srt[sidx];
            srt[sidx] = ((unsigned long long)__float_as_uint(vm_exp(xs[i] - m) / S) << 32) | i;
        }
        __syncthreads();
        // ... so that the ascending walk with the fp64 running sum (ATen's cumsum order) is a bare add/compare per entry.
        // Masked (-inf) entries precede everything and add exactly 0; the removed set is a prefix of the order (cum is
        // non-decreasing); the last (largest) entry is never removed.
        //
        // The walk is sequential by definition (fp64 additions in ascending order, each running sum rounded to fp32 and compared with thr),
        // but its OUTCOME — how long the removed prefix is — can be decided in parallel: any order of summing non-negative doubles is within
        // n * 2^-53 (relative) of any other, so a blocked parallel prefix sum C^ is within 1e-12 of the sequential running sum c, and
        // (float)c <= thr  <=>  c below the midpoint `mid` of thr and the next float (at the midpoint itself: ties-to-even).  Whenever every
        // C^ is farther than 1e-9 (relative) from mid, the parallel decision IS the sequential one; otherwise (and for a tie crowd beyond the
        // sort buffer, and for degenerate thresholds) one thread walks as before.
        bool serial = split || !(thr > 1e-30f) || cnt < 2 || force_walk;
        if (!serial) {
            const double mid = 0.5 * ((double)thr + (double)__uint_as_float(__float_as_uint(thr) + 1u));
            const int g = (cnt + 255) >> 8, s0 = tid * g, s1 = (s0 + g < cnt) ? s0 + g : cnt;
            double loc = 0.0;
            for (int sidx = s0; sidx < s1; ++sidx) loc += (double)__uint_as_float((unsigned)(srt[sidx] >> 32));
            const double base = vs_scan256_d(loc, s_cd) - loc;            // sum of everything before this thread's run
            double c = base; int nrm = 0, edge = 0;
            for (int sidx = s0; sidx < s1; ++sidx) {
                c += (double)__uint_as_float((unsigned)(srt[sidx] >> 32));
                if (sidx < cnt - 1) {                                      // (the last entry is never removed)
                    if (fabs(c - mid) <= 1e-9 * mid) edge = 1;
                    else if (c < mid) ++nrm;
                }
            }
            if (edge) atomicOr(&s_flag, 1);
            __syncthreads();
            serial = s_flag != 0;
            if (!serial)
                for (int sidx = s0; sidx < s0 + nrm; ++sidx) xs[(unsigned)srt[sidx]] = -INFINITY;     // (c is non-decreasing: this thread's removed entries are a prefix of its run)
        }
        // (entries are fetched eight at a time so the LDS latency is paid once per batch, not once per add)
        if (serial && tid == 0) {
            double c = 0.0;
            bool done = false;
            int left = total - 1;                                   // entries that may still be removed
            if (split) {                                            // tie group first, ascending index; all share one probability
                float pT = 0.f;
                for (int i = 0; i < V && left > 0 && !done; ++i) {
                    const float v = xs[i];
                    if (!(v > -INFINITY) || vm_float_key(v) != T) continue;
                    pT = vm_exp(v - m) / S;
                    c += (double)pT;
                    if ((float)c <= thr) { xs[i] = -INFINITY; --left; } else done = true;
                }
            }
            for (int s0 = 0; s0 < cnt && left > 0 && !done; s0 += 8) {
                unsigned long long e[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) e[j] = srt[s0 + j < cnt ? s0 + j : cnt - 1];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (done || left <= 0 || s0 + j >= cnt) continue;
                    c += (double)__uint_as_float((unsigned)(e[j] >> 32));
                    if ((float)c <= thr) { xs[(unsigned)e[j]] = -INFINITY; --left; } else done = true;
                }
            }
        }
        }   // !cut_done
        __syncthreads();
    }

    // (4) softmax of what is left, divide by the Exp(1) noise, arg-max (first max; NaN wins, as torch.argmax)
    float part = 0.f;
    if constexpr (NV > 0) {
        // what top-p removed (and top-k before it) is -inf now: exp(-inf - m) is exactly 0; everything else kept its exponential
#pragma unroll
        for (int k = 0; k < NV; ++k) { const float x = xs[tid + 256 * k]; ev[k] = use_top_p ? (x == -INFINITY ? 0.0f : ev[k]) : vm_exp(x - m); part = part + ev[k]; }
    } else
    for (int i = tid; i < V; i += 256) part = part + vm_exp(xs[i] - m);
    const float S = vh_block_sum256(part, red);
    const float* qn = noise + row * V;
    float bv = 0.f; int bi = -1; int bn = 0;
    auto consider = [&](float rv, int i) {
        const int isn = (rv != rv);
        if (bi < 0) { bv = rv; bi = i; bn = isn; }
        else if (!bn && (isn || rv > bv)) { bv = rv; bi = i; bn = isn; }
    };
    if constexpr (NV > 0) {
#pragma unroll
        for (int k = 0; k < NV; ++k) consider((ev[k] / S) / qv[k], tid + 256 * k);
    } else
    for (int i = tid; i < V; i += 256) consider((vm_exp(xs[i] - m) / S) / qn[i], i);
    // combine across threads: NaN beats number; larger beats smaller; ties -> smaller index
    auto better = [](float av, int ai, int an, float cv, int ci, int cn) -> bool {      // is (c) better than (a)?
        if (ci < 0) return false;
        if (ai < 0) return true;
        if (an != cn) return cn != 0;
        if (an) return ci < ai;
        if (cv != av) return cv > av;
        return ci < ai;
    };
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float ov = __shfl_xor(bv, off, 64); const int oi = __shfl_xor(bi, off, 64); const int on = __shfl_xor(bn, off, 64);
        if (better(bv, bi, bn, ov, oi, on)) { bv = ov; bi = oi; bn = on; }
    }
    if ((tid & 63) == 0) { s_bv[tid >> 6] = bv; s_bi[tid >> 6] = bi; s_bn[tid >> 6] = bn; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; ++w) if (better(bv, bi, bn, s_bv[w], s_bi[w], s_bn[w])) { bv = s_bv[w]; bi = s_bi[w]; bn = s_bn[w]; }
        idx_out[row] = bi;
    }
    if (masked_out) for (int i = tid; i < V; i += 256) masked_out[row * V + i] = xs[i];
}

static int g_sampler_force_walk = 0;
// testing: 1 = decide every top-p cut by the sequential walk (the definition), 0 = by the parallel prefix sum wherever that is provably the same
extern "C" int varhip_sampler_force_walk(int on) { g_sampler_force_walk = on ? 1 : 0; return 0; }

extern "C" int varhip_cfg_sample_f32(const float* logits, const float* noise, int64_t* idx_out, float* masked_out,
                                     int B, int l, int V, double t_cfg, int top_k, double top_p, varhip_stream_t stream) {
    if (B <= 0 || l <= 0 || V <= 0 || (V & 255) || V > 8192 || top_k < 0 || top_k > V) return VARHIP_EINVAL;
    if ((uintptr_t)logits & 15) return VARHIP_EINVAL;                 // (rows are read 16 bytes per lane)
    const int64_t rows = (int64_t)B * l;
    int cap = 2; while (cap < (top_k > 0 ? top_k : V)) cap <<= 1;       // sort buffer entries (see the kernel: ties beyond it are handled unsorted)
    const size_t lds = sizeof(float) * (size_t)V + sizeof(unsigned long long) * (size_t)cap;
    static bool attr_done = false;
    if (!attr_done) { (void)hipFuncSetAttribute((const void*)k_cfg_sample<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 12 * 8192);
                      (void)hipFuncSetAttribute((const void*)k_cfg_sample<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 12 * 8192); attr_done = true; }
    VhScope sc(VH_FAM_SAMPLER, (hipStream_t)stream, 0, 4.0 * rows * V * 3.0);
    if (V == 4096)
        hipLaunchKernelGGL(k_cfg_sample<16>, dim3((unsigned)rows), dim3(256), lds, (hipStream_t)stream, logits, noise, idx_out, masked_out, rows, V,
                           (float)(1.0 + t_cfg), (float)t_cfg, top_k, top_p > 0.0 ? 1 : 0, (float)(1.0 - top_p), cap, g_sampler_force_walk);
    else
        hipLaunchKernelGGL(k_cfg_sample<0>, dim3((unsigned)rows), dim3(256), lds, (hipStream_t)stream, logits, noise, idx_out, masked_out, rows, V,
                           (float)(1.0 + t_cfg), (float)t_cfg, top_k, top_p > 0.0 ? 1 : 0, (float)(1.0 - top_p), cap, g_sampler_force_walk);
    return vh_launch_status();
}

// ---- gumbel softmax of the more_smooth path (helpers.py:22-36, var.py:178-180): y = softmax((x*mul - ln(noise)) / tau) per row.
// One 256-thread workgroup per row; the row sum is the canonical W256 sum, -ln is include/var_math.h's vm_log.
__global__ void __launch_bounds__(256) k_gumbel_softmax(const float* __restrict__ x, const float* __restrict__ noise, float* __restrict__ y,
                                                        int V, float mul, float tau) {
    extern __shared__ __attribute__((aligned(16))) float zs[];
    __shared__ float red[4];
    const int tid = threadIdx.x;
    const int64_t row = blockIdx.x;
    float m = -INFINITY;
    for (int i = tid; i < V; i += 256) {
        const float g = -vm_log(noise[row * V + i]);
        const float z = (x[row * V + i] * mul + g) / tau;
        zs[i] = z; m = fmaxf(m, z);
    }
    m = vh_block_max256(m, red);
    float part = 0.f;
    for (int i = tid; i < V; i += 256) { const float e = vm_exp(zs[i] - m); zs[i] = e; part = part + e; }
    const float S = vh_block_sum256(part, red);
    for (int i = tid; i < V; i += 256) y[row * V + i] = zs[i] / S;
}

extern "C" int varhip_gumbel_softmax_f32(const float* x, const float* noise, float* y, int64_t rows, int V, float mul, float tau, varhip_stream_t stream) {
    if (rows < 0 || V <= 0 || (V & 255) || V > 16384) return VARHIP_EINVAL;
    if (rows == 0) return 0;
    VhScope sc(VH_FAM_SAMPLER, (hipStream_t)stream, 0, 12.0 * rows * V);
    hipLaunchKernelGGL(k_gumbel_softmax, dim3((unsigned)rows), dim3(256), sizeof(float) * V, (hipStream_t)stream, x, noise, y, V, mul, tau);
    return vh_launch_status();
}
