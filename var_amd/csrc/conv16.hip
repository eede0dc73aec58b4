// conv16.hip — the VQVAE decoder's 3x3 convolutions in the 16-bit throughput mode: channels-last fp16 activations and weights, fp32
// accumulation on v_mfma_f32_16x16x32_f16, implicit GEMM (reference basic_vae.py:22-28,40-60,163-226 under the harness' fp16 autocast).
// Not part of the fp32 parity contract (see gemm16.hip); the decoder is off the token path in either mode.
//
// Same scheme as the CONV instantiations of k_dma_gemm (gemm.hip): 128 or 256 output pixels x (32*TNW) output channels per workgroup, the
// input tile read through a buffer descriptor whose window starts one row + one pixel before the first sample the tile touches (a padded
// tap ORs bit 31 into the lane offset: the hardware bounds check writes zeros), weights [Cout][tap][Cin], K order = 32-channel chunk
// outermost, then tap (the taps of a chunk re-read the same cache lines), Upsample2x as four 2x2 phase convolutions.
// Differences forced by the 16x rate of the matrix pipe: a K tile is ONE 32-channel chunk of one tap = 64 bytes per row = one MFMA
// k-step, so the LDS pipeline is NST stages deep with counted s_waitcnt vmcnt (a tile's 20 MFMAs per wave are over in ~320 cycles,
// far less than one memory latency).  LDS rows of 64 bytes: slot c of row r holds chunk c ^ ((-(r >> 2)) & 3) (conflict-free b128 reads).
#include "common.h"
#include "elem16.h"

namespace VH16_NS {

typedef vh_e16 h8 __attribute__((ext_vector_type(8)));
typedef vh_e16 h4 __attribute__((ext_vector_type(4)));

struct Conv16P {
    const vh_e16* in; const vh_e16* w; const float* bias; void* out; const vh_e16* resid; double* gn_part;
    int M, N, K;                   // pixels (of the low-res map in the phase mode), Cout, taps * Cin
    int H, Wd, Cin, phase, out_mode;     // H, Wd: the map the M pixels live on; phase: Upsample2x phase form (grid.z = 4 phases)
    int64_t sW;                    // element stride between the phase weight sets
    int tilesM, tilesN;
    // k_conv16h<.., GN = true>: the input is the RAW map; GroupNorm + SiLU are applied to the halo patch where it lies in LDS
    const float* gn_table; int gn_silu;      // [B][2][Cin]: y = x * table[b][0][c] + table[b][1][c]
};

__device__ __forceinline__ void vh16c_dma_glob(const void* base, uint32_t voff, uint32_t lds) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(base), "s"(lds) : "memory", "m0");
}
__device__ __forceinline__ void vh16c_dma_buf(__amdgpu_buffer_rsrc_t rsrc, uint32_t voff, uint32_t soff, uint32_t lds) {
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds"
                 : : "v"(voff), "s"(rsrc), "s"(__builtin_amdgcn_readfirstlane(soff)), "s"(__builtin_amdgcn_readfirstlane(lds)) : "memory", "m0");
}
template <int N> __device__ __forceinline__ void vh16c_waitcnt_barrier() { asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory"); }
template <int MAXA, int PER> __device__ __forceinline__ void vh16c_wait_dma_and_barrier(int ahead) {      // `ahead` is wave-uniform
    if constexpr (MAXA == 0) vh16c_waitcnt_barrier<0>();
    else { if (ahead >= MAXA) vh16c_waitcnt_barrier<MAXA * PER>(); else vh16c_wait_dma_and_barrier<MAXA - 1, PER>(ahead); }
}

template <int TNW, int NST, int WM, int OCC = 2>
__global__ void __launch_bounds__(WM * 128) __attribute__((amdgpu_waves_per_eu(WM * OCC / 2, WM * OCC / 2))) k_conv16(Conv16P p) {
    // WM x 2 waves, each 64 pixels x (16*TNW) channels, OCC workgroups per CU: WM = 2 -> 128-pixel tiles; WM = 4 -> 256-pixel tiles (the weight
    // tile is fetched from L2 once per 256 pixels: 98 instead of 71 FLOP per byte moved into LDS — that traffic, not the matrix pipe or the
    // LDS reads, bounds the loop: with the loads taken out it runs no faster).  Two workgroups per CU either way: one's epilogue and pipeline
    // fill hide behind the other's loop (one 8-wave workgroup per CU with 5 stages measured 15 % slower than two with 3).
    constexpr int TMW = 4, NWAVE = WM * 2, BM = WM * 64, BN = TNW * 32, ROWB = 64;
    constexpr int NIA = BM / 16 / NWAVE, NBT = BN / 16;           // DMA instructions (16 rows x 64 B) per K tile: NIA per wave for the pixels, NBT per workgroup for the weights
    constexpr int NIB_HI = (NBT + NWAVE - 1) / NWAVE, NIB_LO = NBT / NWAVE, N_HI = NBT - NIB_LO * NWAVE;   // waves < N_HI issue NIB_HI weight loads, the rest NIB_LO
    constexpr int STAGE = (BM + BN) * ROWB, PER_HI = NIA + NIB_HI, PER_LO = NIA + NIB_LO;
    static_assert((NST - 2) * PER_HI <= 63, "vmcnt is a 6-bit counter");
    extern __shared__ __attribute__((aligned(16))) char smc[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const bool hi_wave = wave < N_HI;
    int tm_, tn_;
    {
        const int nwg = p.tilesM * p.tilesN, bid = blockIdx.x;
        const int q = nwg >> 3, rem = nwg & 7, xcd = bid & 7, loc = bid >> 3;
        const int lin = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + loc;
        const int GM = 8, width = GM * p.tilesN, group = lin / width, first = group * GM;
        const int gsz = (p.tilesM - first) < GM ? (p.tilesM - first) : GM;
        tm_ = first + (lin % width) % gsz;
        tn_ = (lin % width) / gsz;
    }
    const int m0 = tm_ * BM, n0 = tn_ * BN, bz = blockIdx.z;
    const int hw = p.H * p.Wd, ntap = p.phase ? 4 : 9;
    const vh_e16* Wb = p.w + (int64_t)bz * p.sW;

    // input window of this workgroup's buffer descriptor: from one row + one pixel before the first sample the tile touches
    const int mlast = (m0 + BM - 1 < p.M ? m0 + BM - 1 : p.M - 1), mfirst = m0 < p.M ? m0 : p.M - 1;
    const int cv_b0 = mfirst / hw;
    const int64_t sample = (int64_t)hw * p.Cin, shift = (int64_t)(p.Wd + 1) * p.Cin;
    const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.in + (int64_t)cv_b0 * sample - shift), 0,
                                                                            (int)((((int64_t)(mlast / hw - cv_b0 + 1)) * sample + shift) * 2), 0x00020000);
    const int drow = lane >> 2, dslot = lane & 3, dchunk = dslot ^ ((-(drow >> 2)) & 3);
    uint32_t aoff[NIA], abad[NIA], boff[NIB_HI];
    const int dy0 = p.phase ? (bz >> 1) - 1 : -1, dx0 = p.phase ? (bz & 1) - 1 : -1;
#pragma unroll
    for (int i = 0; i < NIA; ++i) {
        int m = m0 + (wave * NIA + i) * 16 + drow; m = m < p.M ? m : p.M - 1;
        const int b = m / hw, rem2 = m - b * hw, y = rem2 / p.Wd, x = rem2 - y * p.Wd;
        aoff[i] = (uint32_t)(((((int64_t)(b - cv_b0) * p.H + y) * p.Wd + x) * p.Cin + dchunk * 8) * 2);
        const bool r0 = (unsigned)(y + dy0) >= (unsigned)p.H, r1 = (unsigned)(y + dy0 + 1) >= (unsigned)p.H, r2 = (unsigned)(y + dy0 + 2) >= (unsigned)p.H;
        const bool c0 = (unsigned)(x + dx0) >= (unsigned)p.Wd, c1 = (unsigned)(x + dx0 + 1) >= (unsigned)p.Wd, c2 = (unsigned)(x + dx0 + 2) >= (unsigned)p.Wd;
        abad[i] = p.phase ? ((r0 ? 0x3u : 0u) | (r1 ? 0xCu : 0u) | (c0 ? 0x5u : 0u) | (c1 ? 0xAu : 0u))
                          : ((r0 ? 0x007u : 0u) | (r1 ? 0x038u : 0u) | (r2 ? 0x1C0u : 0u) | (c0 ? 0x049u : 0u) | (c1 ? 0x092u : 0u) | (c2 ? 0x124u : 0u));
    }
#pragma unroll
    for (int i = 0; i < NIB_HI; ++i) {                            // weight load i of this wave covers rows 16*(wave + NWAVE*i) .. +15 of the tile
        int n = n0 + (wave + NWAVE * i) * 16 + drow; n = n < p.N ? n : p.N - 1;
        boff[i] = (uint32_t)(((int64_t)n * p.K + dchunk * 8) * 2);
    }
    int cv_tap = 0, cv_cc = 0;                                    // tap / channel chunk of the next K tile to be requested (tiles are requested in order)
    auto dma_tile = [&](int st) {
        char* sA = smc + st * STAGE + wave * NIA * 16 * ROWB;
        char* sB = smc + st * STAGE + BM * ROWB + wave * 16 * ROWB;
        const int tap = cv_tap, ci0 = cv_cc * 32;
        if (++cv_tap == ntap) { cv_tap = 0; ++cv_cc; }
        int dy, dx;
        if (p.phase) { dy = (tap >> 1) + dy0; dx = (tap & 1) + dx0; }
        else { const int ky = tap / 3; dy = ky - 1; dx = tap - ky * 3 - 1; }
        const uint32_t soff = (uint32_t)(((dy + 1) * p.Wd + dx + 1) * p.Cin + ci0) * 2u;
#pragma unroll
        for (int i = 0; i < NIA; ++i)
            vh16c_dma_buf(arsrc, ((abad[i] >> tap) << 31) | aoff[i], soff, (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)(sA + i * 16 * ROWB));
#pragma unroll
        for (int i = 0; i < NIB_HI; ++i)
            if (i < NIB_LO || hi_wave)
                vh16c_dma_glob((const char*)Wb + (size_t)(tap * p.Cin + ci0) * 2, boff[i], (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)(sB + i * NWAVE * 16 * ROWB));
    };
    auto wait_tile = [&](int ahead) {                             // this wave's requests for the tile about to be read have landed; then everyone's
        if (PER_HI != PER_LO && hi_wave) vh16c_wait_dma_and_barrier<NST - 2, PER_HI>(ahead);
        else vh16c_wait_dma_and_barrier<NST - 2, PER_LO>(ahead);
    };

    f32x4 acc[TMW][TNW];
#pragma unroll
    for (int i = 0; i < TMW; ++i)
#pragma unroll
        for (int j = 0; j < TNW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int r16 = lane & 15, kq = lane >> 4;
    const int nk = p.K / 32;
    const int rsl = (kq ^ ((-(r16 >> 2)) & 3)) << 4;              // the lane's 16-byte slot inside a 64-byte LDS row
    const bool idle_wave = (n0 + wn * TNW * 16 >= p.N) || (m0 + wm * TMW * 16 >= p.M);
#pragma unroll
    for (int t = 0; t < NST - 1; ++t) if (t < nk) dma_tile(t);
    if (idle_wave) {                                              // same requests and barriers, no matrix work (kept out of the hot loop)
        for (int kt = 0; kt < nk; ++kt) {
            const int ahead = (nk - 1 - kt) < (NST - 2) ? (nk - 1 - kt) : (NST - 2);
            wait_tile(ahead);
            if (kt + NST - 1 < nk) dma_tile((kt + NST - 1) % NST);
        }
    } else {
        for (int kt = 0; kt < nk; ++kt) {
            const int ahead = (nk - 1 - kt) < (NST - 2) ? (nk - 1 - kt) : (NST - 2);
            wait_tile(ahead);
            const int cur = kt % NST;
            const char* sA = smc + cur * STAGE + (wm * TMW * 16 + r16) * ROWB + rsl;
            const char* sB = smc + cur * STAGE + BM * ROWB + (wn * TNW * 16 + r16) * ROWB + rsl;
            h8 am[TMW], bn[TNW];
#pragma unroll
            for (int i = 0; i < TMW; ++i) am[i] = *(const h8*)(sA + i * 16 * ROWB);
#pragma unroll
            for (int j = 0; j < TNW; ++j) bn[j] = *(const h8*)(sB + j * 16 * ROWB);
            if (kt + NST - 1 < nk) dma_tile((kt + NST - 1) % NST);
#pragma unroll
            for (int i = 0; i < TMW; ++i)
#pragma unroll
                for (int j = 0; j < TNW; ++j) {
                    // (>= 16 accumulators: pinned in place, elem16.h — an accumulator's next MFMA is a whole K tile behind; the small tiles keep the builtin)
                    if constexpr (TMW * TNW >= 16) VH16_MFMA_16x16x32_INPLACE(bn[j], am[i], acc[i][j]);
                    else acc[i][j] = VH16_MFMA_16x16x32(bn[j], am[i], acc[i][j]);
                }
        }
    }
    __syncthreads();                                              // the stages are free: the GroupNorm partials below reuse them

    // ---- epilogue: acc[i][j][e] = C[pixel m = tile_m(i) + r16][channel n = tile_n(j) + 4*kq + e]
    const int nw0 = n0 + wn * TNW * 16;
    const bool vec = !(p.N & 3);
    constexpr int SROW = TNW * 64 + 16, STG = 16 * SROW;          // fp32 staging: 16 pixels x (16*TNW) channels per wave, rows padded by 16 B (conflict-free b128 writes)
    double* const red = reinterpret_cast<double*>(smc + NWAVE * STG);
    static_assert((size_t)NWAVE * STG + (size_t)WM * BN * 2 * sizeof(double) <= (size_t)NST * STAGE, "epilogue staging must fit in the stages");
    if (vec && p.out_mode == 0) {
        // Through LDS, 16 pixels at a time and private to the wave (no workgroup barrier): the accumulator layout gives a lane 4 channels of
        // one pixel, so direct stores are 32-byte pieces of 16 different rows per instruction — measured at a third of the kernel's time on
        // the 160-channel maps.  Read back, NC = 4*TNW consecutive lanes cover one pixel's 32*TNW bytes: the residual read and the store are
        // contiguous runs, and each lane keeps ONE 4-channel column over all its pixels, which is what the GroupNorm partial needs.
        constexpr int NC = TNW * 4, PXI = 64 / NC, NIT = (16 + PXI - 1) / PXI;
        char* const stg = smc + wave * STG;
        const int col = lane % NC, pl = lane / NC;
        const int n = nw0 + col * 4;
        const bool lane_on = pl < PXI && n < p.N;
        float b4[TNW][4];
#pragma unroll
        for (int j = 0; j < TNW; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) { const int nn = nw0 + j * 16 + kq * 4 + e; b4[j][e] = nn < p.N ? p.bias[nn] : 0.f; }
        float gs[4] = {0.f, 0.f, 0.f, 0.f}, gq[4] = {0.f, 0.f, 0.f, 0.f};      // fp32 over this lane's <= 64 / PXI pixels (values are fp16: 11 bits), fp64 from there on
#pragma unroll
        for (int i = 0; i < TMW; ++i) {
            const int mrow = m0 + (wm * TMW + i) * 16;
            if (mrow >= p.M) break;                                           // wave-uniform
#pragma unroll
            for (int j = 0; j < TNW; ++j) {
                f32x4 v = acc[i][j];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = v[e] + b4[j][e];
                *(f32x4*)(stg + r16 * SROW + (j * 16 + kq * 4) * 4) = v;
            }
            int mo16 = 0;                                                     // phase mode: output row of pixel (lane & 15) of this pass, handed to its reader below
            if (p.phase) {
                int m = mrow + r16; m = m < p.M ? m : p.M - 1;
                const int b = m / hw, rem2 = m - b * hw, y = rem2 / p.Wd, x = rem2 - y * p.Wd;
                mo16 = (b * (2 * p.H) + 2 * y + (bz >> 1)) * (2 * p.Wd) + 2 * x + (bz & 1);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                // the wave's own LDS traffic is in order; this keeps the compiler from moving the reads up
#pragma unroll
            for (int k = 0; k < NIT; ++k) {
                const int px = pl + PXI * k, m = mrow + px;
                const int mo_ph = __shfl(mo16, px & 15, 64);
                if (!lane_on || px >= 16 || m >= p.M) continue;
                f32x4 v = *(const f32x4*)(stg + px * SROW + col * 16);
                if (p.resid) { const h4 r4 = *(const h4*)(p.resid + (int64_t)m * p.N + n);
#pragma unroll
                               for (int e = 0; e < 4; ++e) v[e] = (float)r4[e] + v[e]; }
                h4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (vh_e16)v[e];
                *(h4*)((vh_e16*)p.out + (int64_t)(p.phase ? mo_ph : m) * p.N + n) = o;
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float d = (float)o[e]; gs[e] += d; gq[e] += d * d; }     // statistics of what the next GroupNorm will read: the rounded values
            }
            asm volatile("" ::: "memory");
        }
        if (p.gn_part) {                                                      // host guarantees full tiles; lanes pl = 0..PXI-1 hold the same column
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float s = gs[e], q = gq[e];
#pragma unroll
                for (int t = 1; t < PXI; ++t) { s += __shfl(gs[e], lane + NC * t, 64); q += __shfl(gq[e], lane + NC * t, 64); }
                if (pl == 0 && n < p.N) { const int nl = wn * TNW * 16 + col * 4 + e; red[(wm * BN + nl) * 2] = (double)s; red[(wm * BN + nl) * 2 + 1] = (double)q; }
            }
        }
    } else if (nw0 < p.N) {                                                   // the decoder's last conv (fp32 NCHW, <= 3 channels) and channel counts that are not multiples of 4
#pragma unroll
        for (int j = 0; j < TNW; ++j) {
            const int n = nw0 + j * 16 + kq * 4;
            if (n >= p.N) continue;
            float b4[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) b4[e] = (n + e < p.N) ? p.bias[n + e] : 0.f;
#pragma unroll
            for (int i = 0; i < TMW; ++i) {
                const int m = m0 + (wm * TMW + i) * 16 + r16;
                if (m >= p.M) continue;
                const int b = m / hw, rem2 = m - b * hw;
                f32x4 v = acc[i][j];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = v[e] + b4[e];
                if (p.out_mode != 0) {                                      // fp32 NCHW store, clamp (+ de-normalise)
                    for (int e = 0; e < 4; ++e) {
                        if (n + e >= p.N) break;
                        const float x = vm_min(vm_max(v[e], -1.0f), 1.0f);
                        ((float*)p.out)[((int64_t)b * p.N + n + e) * hw + rem2] = p.out_mode == 1 ? (x + 1.0f) * 0.5f : x;
                    }
                    continue;
                }
                int64_t mo = m;                                             // output row; the phase mode scatters to the 2x grid
                if (p.phase) { const int y = rem2 / p.Wd, x = rem2 - y * p.Wd; mo = ((int64_t)b * (2 * p.H) + 2 * y + (bz >> 1)) * (2 * p.Wd) + 2 * x + (bz & 1); }
                for (int e = 0; e < 4; ++e) {
                    if (n + e >= p.N) break;
                    float x = v[e];
                    if (p.resid) x = (float)p.resid[(int64_t)m * p.N + n + e] + x;
                    ((vh_e16*)p.out)[mo * p.N + n + e] = (vh_e16)x;
                }
            }
        }
    }
    if (p.gn_part) {
        __syncthreads();
        if (tid < BN) {                                             // one partial per 128-pixel block (varhip_conv_gn_blocks), WM / 2 of them per tile
#pragma unroll
            for (int hb = 0; hb < WM / 2; ++hb) {
                const int mh = m0 + hb * 128;
                if (mh >= p.M) break;
                const double s = red[(2 * hb * BN + tid) * 2] + red[((2 * hb + 1) * BN + tid) * 2];
                const double q = red[(2 * hb * BN + tid) * 2 + 1] + red[((2 * hb + 1) * BN + tid) * 2 + 1];
                const int b = mh / hw, per = hw / 128, blk = bz * per + (mh - b * hw) / 128, nblk = (int)gridDim.z * per;
                double* o = p.gn_part + (((int64_t)b * nblk + blk) * p.N + n0 + tid) * 2;
                o[0] = s; o[1] = q;
            }
        }
    }
}

// ---- k_conv16h: the same convolution with the INPUT tile fetched once per 32-channel chunk instead of once per tap.
// k_conv16's loop is bound by the bytes it moves from L2 into LDS (98 FLOP per byte at best; with the loads taken out it runs no
// faster), and 62 % of those bytes are the nine shifted copies of the same input pixels.  Here a workgroup owns a PH x PW patch of one
// image (256 pixels); per chunk, the (PH+2) x (PW+2) halo patch goes to LDS once (out-of-image pixels arrive as zeros through the
// descriptor's bounds check) and the nine taps read it at nine row offsets: 112 instead of 234 KiB-pieces per chunk for 160 output channels.
// Roles: waves 0..NBW-1 stream the weights (a [BN][32] tile per tap, three stages, two pieces per wave and step, s_waitcnt vmcnt(2)),
// the other waves fetch the next chunk's patch during taps 0..ASTEPS-1 of the current one (two pieces per step, double-buffered) —
// every wave's wait count is then a constant.  LDS rows of 64 bytes: slot c of row r holds chunk c ^ (((r >> 2) & 1) << 1), the one
// family of swizzles under which ds_read_b128 of 16 CONSECUTIVE rows is conflict-free at ANY starting row (the taps shift it).
// GN = true (round 4; reference basic_vae.py:57-60 `conv(swish(norm(x)))`): the convolution takes the RAW map and the GroupNorm statistics and
// applies y = SiLU(x * sc + sh) — k_gn16_apply's arithmetic operation for operation, one rounding to the 16-bit type — to the next chunk's halo patch
// in place in LDS: the patch is waited for at tap 6 instead of tap 0 of its chunk, every thread then normalises three 16-byte units of it behind the
// MFMAs of taps 6 and 7 (the (scale, shift) of all Cin channels sit in LDS; out-of-image pixels stay the zeros the bounds check wrote: the convolution
// pads the NORMALISED map).  What it removes is the apply pass over the map (read + write of every activation in HBM) in front of every such conv.
// The result is bit-identical to varhip_gn_apply_* followed by the plain convolution (tests).  Needs the registers the in-place accumulators freed.
template <int TNW, int PW, bool GN>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) k_conv16h(Conv16P p) {
    constexpr int TMW = 4, BN = TNW * 32, ROWB = 64;
    constexpr int PH = 256 / PW, P = PW + 2, PROWS = (PH + 2) * P, NPIECE = (PROWS + 15) / 16, PATCH = NPIECE * 1024;
    constexpr int NBT = BN / 16, NBW = NBT / 2, NAW = 8 - NBW, KP = (NPIECE + NAW - 1) / NAW, ASTEPS = (KP + 1) / 2;
    constexpr int BST = BN * ROWB;
    static_assert(NBT % 2 == 0 && NAW >= 1 && ASTEPS <= 6, "roles");
    extern __shared__ __attribute__((aligned(16))) char smc[];
    char* const sB = smc + 2 * PATCH;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const bool b_wave = wave < NBW;
    int tm_, tn_;
    {
        const int nwg = p.tilesM * p.tilesN, bid = blockIdx.x;
        const int q = nwg >> 3, rem = nwg & 7, xcd = bid & 7, loc = bid >> 3;
        const int lin = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + loc;      // an XCD works on consecutive tiles: neighbours share halo rows, and one weight slice
        tn_ = lin / p.tilesM; tm_ = lin - tn_ * p.tilesM;
    }
    const int tX = p.Wd / PW, tY = p.H / PH, tps = tX * tY;
    const int b = tm_ / tps, trem = tm_ - b * tps, tyi = trem / tX, ty0 = tyi * PH, tx0 = (trem - tyi * tX) * PW;
    const int n0 = tn_ * BN, hw = p.H * p.Wd;
    const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.in + (int64_t)b * hw * p.Cin), 0, (int)((int64_t)hw * p.Cin * 2), 0x00020000);
    const int drow = lane >> 2, dchunk = (lane & 3) ^ (((lane >> 4) & 1) << 1);

    // the lane's part of a weight request — row (lane >> 2) of the 16-row piece, 16-byte slot (lane & 3) ^ swizzle — is rebuilt from the lane id at
    // every request (eight vector instructions, volatile: neither hoisted nor kept): as a loop-invariant register it was spilled, and the scratch
    // reload in front of the request brought an `s_waitcnt vmcnt(0)` that drained the weight pipeline at every step
    const char* const wrow = (const char*)p.w + (size_t)(n0 + wave * 32) * p.K * 2;
    const uint32_t k2 = (uint32_t)p.K * 2u;
    auto dma_b = [&](int c, int tap, int st) {                    // weight tile of (chunk c, tap) -> stage st
        uint32_t l;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
        const uint32_t voff = (l >> 2) * k2 + ((((l & 3u) ^ ((l >> 3) & 2u))) << 4);
#pragma unroll
        for (int i = 0; i < 2; ++i)
            vh16c_dma_glob(wrow + (size_t)(tap * p.Cin + c * 32) * 2 + (size_t)i * 16 * p.K * 2, voff,
                           (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)(sB + st * BST + (wave * 2 + i) * 1024));
    };
    // source offset of every patch ROW (its pixel's channel 0 inside the sample, in bytes; bit 31: outside the image or past the patch — the
    // descriptor's bounds check then writes zeros) -> LDS table, once: the tile's geometry costs a dozen VALU and a 64-bit temporary per row, and
    // the loop below keeps no per-lane register for it.  A lane adds its 16-byte slot (bits 4-5; rows are multiples of 64 bytes apart).
    uint32_t* const atab = reinterpret_cast<uint32_t*>(smc + 2 * PATCH + 3 * BST);
    for (int pr = tid; pr < NPIECE * 16; pr += 512) {
        const int py = pr / P, px = pr - py * P, y = ty0 - 1 + py, x = tx0 - 1 + px;
        const bool bad = pr >= PROWS || (unsigned)y >= (unsigned)p.H || (unsigned)x >= (unsigned)p.Wd;
        atab[pr] = bad ? 0x80000000u : (uint32_t)((y * p.Wd + x) * p.Cin * 2);
    }
    float* const scsh = reinterpret_cast<float*>(atab + NPIECE * 16);          // GN: [Cin] scale, then [Cin] shift of this sample
    if constexpr (GN) {
        // the sample's (scale, shift) table comes ready-made (varhip_gn_scale_shift_f32): built here from the statistics it was a chain of two
        // dependent global loads + a division per channel at the head of every tile
        const f32x4* const tab = reinterpret_cast<const f32x4*>(p.gn_table + (int64_t)b * 2 * p.Cin);
        for (int q = tid; q < p.Cin / 2; q += 512) *(f32x4*)(scsh + 4 * q) = tab[q];
    }
    __syncthreads();
    // the lane's table slot is rebuilt from the lane id at every use (two v_mbcnt + a few shifts; volatile, so that it is neither hoisted nor
    // kept): as a loop-invariant register it was the value the allocator spilled, and a scratch reload in front of every patch request brings
    // an `s_waitcnt vmcnt(0)` that drains the request pipeline
    const uint32_t atab_s = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)atab;
    auto dma_a = [&](int c, int k) {                              // piece k of this wave's share of chunk c's patch
        const int j = (wave - NBW) + NAW * k;
        if (k >= KP || j >= NPIECE) return;                       // wave-uniform
        uint32_t l;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
        const uint32_t rowoff = *(const __attribute__((address_space(3))) uint32_t*)(uintptr_t)(atab_s + (uint32_t)(16 * j) * 4u + (l >> 2) * 4u);
        const uint32_t voff = rowoff | (((l & 3u) ^ ((l >> 3) & 2u)) << 4);
        vh16c_dma_buf(arsrc, voff, (uint32_t)(c * 64), (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)(smc + (c & 1) * PATCH + j * 1024));
    };
    // GN: unit `base + lane` of chunk c's patch (a unit = 16 bytes = 8 channels of patch row unit >> 2), normalised in place
    auto norm_patch = [&](int c, int base) {
        uint32_t l;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
        const uint32_t u = (uint32_t)base + l, row = u >> 2;
        if (u >= (uint32_t)(NPIECE * 64) || (atab[row] >> 31)) return;          // past the patch / outside the image: stays zero
        const uint32_t cg = (u & 3u) ^ ((row >> 1) & 2u);                       // the 8-channel group this slot of this row holds (the read swizzle)
        char* const ptr = smc + (c & 1) * PATCH + u * 16u;
        const h8 v = *(const h8*)ptr;
        const float* const sp = scsh + c * 32 + cg * 8;
        const f32x4 s0 = *(const f32x4*)sp, s1 = *(const f32x4*)(sp + 4), h0 = *(const f32x4*)(sp + p.Cin), h1 = *(const f32x4*)(sp + p.Cin + 4);
        float z[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float y = __builtin_fmaf((float)v[e], e < 4 ? s0[e & 3] : s1[e & 3], e < 4 ? h0[e & 3] : h1[e & 3]);
            z[e] = p.gn_silu ? y * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * y)) : y;
        }
        // (the product rounds to fp32 BEFORE the conversion, as in k_gn16_apply: hipcc otherwise merges the two into v_fma_mixlo_f16)
        asm volatile("" : "+v"(z[0]), "+v"(z[1]), "+v"(z[2]), "+v"(z[3]), "+v"(z[4]), "+v"(z[5]), "+v"(z[6]), "+v"(z[7]));
        h8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (vh_e16)z[e];
        *(h8*)ptr = o;
    };
    constexpr int NUNIT = (NPIECE * 64 + 511) / 512;              // units per thread and chunk (3)
    // In the loop the next chunk's patch is waited for at tap 6 (tap 0 of its own chunk without GN) and every thread normalises its NUNIT units behind
    // the MFMAs of taps 6 and 7.  What that costs is the units' vector instructions at full price (~200 per wave and chunk, two transcendentals per
    // element: +12 % on a 160 -> 160 convolution at 256^2, +15 % at 320 channels) — a workgroup's step is a serial chain (barrier, requests, fragment
    // reads, MFMA issue) that the matrix pipe only half fills, so nothing of a wave's own stream hides.  Measured alternatives, all slower
    // (profiles/r04_gnconv_schedules.txt): the units cut into stages between the MFMA groups (+16 %), half of the waves per tap over taps 3 .. 8 with
    // counted waits on the patch pieces (+16 %).  Against the apply pass it replaces (0.51 ms per 256^2 x 160 map at the HBM roof) the fusion keeps
    // about half: 2.34 vs 2.61 ms per 160 -> 160 convolution at 256^2, 0.93 vs 1.08 at 320 -> 160 / 128^2, break-even at 64^2.
    f32x4 acc[TMW][TNW];
#pragma unroll
    for (int i = 0; i < TMW; ++i)
#pragma unroll
        for (int j = 0; j < TNW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int r16 = lane & 15, kq = lane >> 4;
    // byte offset (inside a patch buffer) of this lane's 16 bytes of fragment 0 at tap (0, 0), before the bank swizzle: row * 64 + kq * 16.  A tap /
    // fragment adds a multiple of 64 (rows), so row bit 2 — what the swizzle looks at — is bit 8 of the sum: address = w ^ ((w >> 3) & 32)
    uint32_t ub = (uint32_t)(((PW == 32 ? wm * 2 * P : wm * 4 * P) + r16) * ROWB + (kq << 4));
    const int bsl = (wn * TNW * 16 + r16) * ROWB + ((kq << 4) ^ (((r16 >> 2) & 1) << 5));
    const int nch = p.Cin / 32;

    if (b_wave) { dma_b(0, 0, 0); dma_b(0, 1, 1); }
    else {
#pragma unroll
        for (int k = 0; k < KP; ++k) dma_a(0, k);
    }
    if constexpr (GN) {                                           // chunk 0's patch: landed, seen by everyone, normalised (the loop's first barrier publishes it)
        if (b_wave) asm volatile("s_barrier" ::: "memory"); else vh16c_waitcnt_barrier<0>();
#pragma unroll
        for (int k = 0; k < NUNIT; ++k) norm_patch(0, k * 512 + wave * 64);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    for (int c = 0; c < nch; ++c) {
        const uint32_t pa = (uint32_t)((c & 1) * PATCH);
        asm volatile("" : "+v"(ub));                                          // keeps the 36 tap addresses from being hoisted out of the chunk loop (they would spill)
        // "is there a next chunk" as a value the optimiser cannot see through: with the plain comparison it peels the last chunk into a second copy
        // of the nine taps, and in that copy it renames accumulators between MFMAs (dst != src C) and spills them (43 registers of scratch).
        // The weight waves' schedule is the same in every chunk, the last included: its steps 7 and 8 request the first two weight tiles of chunk 0
        // again (2 of 47 tiles; they land in stages nobody reads any more and are waited for before the epilogue reuses the LDS), so every step
        // waits with the same vmcnt(2).
        int more = __builtin_amdgcn_readfirstlane((c + 1 < nch) ? 1 : 0);
        asm volatile("" : "+s"(more));
        const int cn = more ? c + 1 : 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            if (b_wave) vh16c_waitcnt_barrier<2>();                                                       // this step's weight tile (and whatever is older) has landed
            else if (t == (GN ? 9 - NUNIT : 0)) vh16c_waitcnt_barrier<0>();                                       // this wave's share of the (next) patch has landed
            else asm volatile("s_barrier" ::: "memory");
            if (b_wave) {                                                                               // two steps ahead; stage (t + 2) % 3 was read last in the previous step
                if (t < 7) dma_b(c, t + 2, (t + 2) % 3);
                else dma_b(cn, t - 7, (t + 2) % 3);
            } else if (t < ASTEPS && more) { dma_a(c + 1, 2 * t); dma_a(c + 1, 2 * t + 1); }             // the other patch buffer was read last in the previous chunk
            const int ky = t / 3, kx = t - ky * 3;
            const char* const sb = sB + (t % 3) * BST + bsl;
            // rolling fragments: all four of the pixels', two of the weight tile's at a time (24 registers; all of the weights' + two of the pixels'
            // were 28, and at 128 registers with 80 accumulators those four decide whether the allocator spills inside this loop)
            h8 am[TMW], bq[2];
            // (volatile add: rows of different (fragment, tap) pairs coincide — fragment 2 at ky is fragment 0 at ky + 1 — and the compiler kept
            // such addresses in registers from one tap to the other: spills inside this loop.  Three vector instructions per read instead.)
            auto lda = [&](int i) { const uint32_t off = pa + (uint32_t)(((PW == 32 ? (i >> 1) * P + (i & 1) * 16 : i * P) + ky * P + kx) * ROWB);
                                    uint32_t w;
                                    asm volatile("v_add_u32 %0, %1, %2" : "=v"(w) : "v"(ub), "s"(off));
                                    return *(const h8*)(smc + (w ^ ((w >> 3) & 32u))); };
            am[0] = lda(0);
            bq[0] = *(const h8*)(sb);
            am[1] = lda(1);
            if (TNW > 1) bq[1] = *(const h8*)(sb + 16 * ROWB);
#pragma unroll
            for (int i = 2; i < TMW; ++i) am[i] = lda(i);
#pragma unroll
            for (int j = 0; j < TNW; ++j) {
#pragma unroll
                for (int i = 0; i < TMW; ++i) VH16_MFMA_16x16x32_INPLACE(bq[j & 1], am[i], acc[i][j]);
                __builtin_amdgcn_sched_barrier(0);
                if (j + 2 < TNW) bq[j & 1] = *(const h8*)(sb + (j + 2) * 16 * ROWB);
            }
            if constexpr (GN) {
                if (t == 9 - NUNIT && more) {
#pragma unroll
                    for (int k = 0; k < NUNIT - 1; ++k) norm_patch(c + 1, k * 512 + wave * 64);
                }
                if (t == 10 - NUNIT && more) { norm_patch(c + 1, (NUNIT - 1) * 512 + wave * 64); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }   // drained before the barrier that opens the next chunk
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);                            // nothing of the epilogue moves up into the last taps (it spilled accumulators there)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // (the weight waves' two surplus requests)
    __syncthreads();                                              // patches and stages are free: the epilogue reuses them
    asm volatile("" ::: "memory");

    // ---- epilogue (as k_conv16's staged one; tiles are always full here)
    constexpr int SROW = TNW * 64 + 16, STG = 16 * SROW, NC = TNW * 4, PXI = 64 / NC, NIT = (16 + PXI - 1) / PXI;
    static_assert((size_t)8 * STG + (size_t)4 * BN * 2 * sizeof(double) <= (size_t)2 * PATCH + 3 * BST, "epilogue staging must fit");
    double* const red = reinterpret_cast<double*>(smc + 8 * STG);
    char* const stg = smc + wave * STG;
    // the epilogue's per-lane indices come from a lane id read HERE: derived from the `lane` of the prologue they were registers the allocator
    // kept (spilled) across the whole K loop
    int lane_e;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));
    const int r16e = lane_e & 15, kqe = lane_e >> 4, tid_e = wave * 64 + lane_e;
    const int nw0 = n0 + wn * TNW * 16, col = lane_e % NC, pl = lane_e / NC, n = nw0 + col * 4;
    const bool lane_on = pl < PXI;
    const f32x4 bcol = lane_on ? *(const f32x4*)(p.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};      // added after the transposition (see k_conv16): 4 registers, not 4 * TNW
    float gs[4] = {0.f, 0.f, 0.f, 0.f}, gq[4] = {0.f, 0.f, 0.f, 0.f};
    // the residual rows of a pass are requested one pass ahead (the first before the first pass parks): fetched inside the pass, each of the
    // four passes waited out a memory latency with the matrix pipe idle — 12-14 % of a 160-channel convolution with a residual
    auto pass_row = [&](int i) -> int64_t {
        const int trow = PW == 32 ? wm * 2 + (i >> 1) : wm * 4 + i, x0 = PW == 32 ? (i & 1) * 16 : 0;
        return ((int64_t)b * p.H + ty0 + trow) * p.Wd + tx0 + x0;
    };
    h4 rcur[NIT], rnxt[NIT];
    auto load_res = [&](int i, h4* r) {
        const int64_t mrow = pass_row(i);
#pragma unroll
        for (int k = 0; k < NIT; ++k) { const int px = pl + PXI * k; if (lane_on && px < 16) r[k] = *(const h4*)(p.resid + (mrow + px) * p.N + n); }
    };
    if (p.resid) load_res(0, rcur);
#pragma unroll
    for (int i = 0; i < TMW; ++i) {
        const int64_t mrow = pass_row(i);
#pragma unroll
        for (int j = 0; j < TNW; ++j) *(f32x4*)(stg + r16e * SROW + (j * 16 + kqe * 4) * 4) = acc[i][j];
        if (p.resid && i + 1 < TMW) load_res(i + 1, rnxt);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int px = pl + PXI * k;
            if (!lane_on || px >= 16) continue;
            const int64_t m = mrow + px;
            f32x4 v = *(const f32x4*)(stg + px * SROW + col * 16) + bcol;
            if (p.resid) { const h4 r4 = rcur[k];
#pragma unroll
                           for (int e = 0; e < 4; ++e) v[e] = (float)r4[e] + v[e]; }
            h4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (vh_e16)v[e];
            *(h4*)((vh_e16*)p.out + m * p.N + n) = o;
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = (float)o[e]; gs[e] += d; gq[e] += d * d; }
        }
#pragma unroll
        for (int k = 0; k < NIT; ++k) rcur[k] = rnxt[k];
        asm volatile("" ::: "memory");
    }
    if (p.gn_part) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float sm = gs[e], q = gq[e];
#pragma unroll
            for (int t = 1; t < PXI; ++t) { sm += __shfl(gs[e], lane_e + NC * t, 64); q += __shfl(gq[e], lane_e + NC * t, 64); }
            if (pl == 0) { const int nl = wn * TNW * 16 + col * 4 + e; red[(wm * BN + nl) * 2] = (double)sm; red[(wm * BN + nl) * 2 + 1] = (double)q; }
        }
        __syncthreads();
        if (tid_e < BN) {                                           // two partials per tile (its upper and lower 128 pixels); any partition of a sample's pixels serves the statistics
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) {
                const double sm = red[(2 * hb * BN + tid_e) * 2] + red[((2 * hb + 1) * BN + tid_e) * 2];
                const double q = red[(2 * hb * BN + tid_e) * 2 + 1] + red[((2 * hb + 1) * BN + tid_e) * 2 + 1];
                double* o = p.gn_part + (((int64_t)b * (2 * tps) + 2 * trem + hb) * p.N + n0 + tid_e) * 2;
                o[0] = sm; o[1] = q;
            }
        }
    }
}

// LDS of one k_conv16h workgroup: two patch buffers, three weight stages, the row table (+ the GroupNorm (scale, shift) table); two workgroups per CU
template <int TNW, int PW>
static constexpr size_t conv16h_lds(int gn_cin) {
    return (size_t)2 * ((((256 / PW) + 2) * (PW + 2) + 15) / 16) * 1024 + (size_t)3 * TNW * 32 * 64 + (size_t)((((256 / PW) + 2) * (PW + 2) + 15) / 16) * 64 + (size_t)gn_cin * 8;
}
template <int TNW, int PW, bool GN>
static int launch_conv16h(Conv16P& p, hipStream_t s) {
    constexpr int BN = TNW * 32, PH = 256 / PW;
    static_assert(conv16h_lds<TNW, PW>(0) <= 80 * 1024, "two workgroups per CU");
    const size_t lds = conv16h_lds<TNW, PW>(GN ? p.Cin : 0);
    if (lds > 80 * 1024) return VARHIP_EINVAL;
    p.tilesM = (p.M / (p.H * p.Wd)) * (p.H / PH) * (p.Wd / PW); p.tilesN = p.N / BN;
    auto kfn = k_conv16h<TNW, PW, GN>;
    static bool attr_done = false;
    if (!attr_done) { (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024); attr_done = true; }
    hipLaunchKernelGGL(kfn, dim3(p.tilesM * p.tilesN), dim3(512), lds, s, p);
    return vh_launch_status();
}

template <int TNW, int NST, int WM, int OCC = 2>
static int launch_conv16(Conv16P& p, int nz, hipStream_t s) {
    constexpr int BN = TNW * 32, BM = WM * 64;
    constexpr size_t lds = (size_t)NST * (BM + BN) * 64;
    static_assert(lds <= 160 * 1024 / OCC, "LDS budget of the intended occupancy");
    p.tilesM = (p.M + BM - 1) / BM; p.tilesN = (p.N + BN - 1) / BN;
    auto kfn = k_conv16<TNW, NST, WM, OCC>;
    static bool attr_done = false;
    if (!attr_done) { if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_done = true; }
    hipLaunchKernelGGL(kfn, dim3(p.tilesM * p.tilesN, 1, nz), dim3(WM * 128), lds, s, p);
    return vh_launch_status();
}
// which kernel takes a launch: 1 / 2 = the halo-patch kernel with 8x32 / 16x16 patches, 3 = 256-pixel tiles, 0 = 128-pixel tiles
static int pick_conv16(const Conv16P& p, int nz) {
    // the halo-patch kernel: plain 3x3 convs on maps that tile into 8x32 or 16x16 patches, once there is a workgroup for every CU
    if (!p.phase && p.out_mode == 0 && vh_g_conv16_force_wm != 2 && vh_g_conv16_force_wm != 4 && (p.N % 160 == 0 || p.N % 128 == 0)
        && (int64_t)p.H * p.Wd * p.Cin * 2 < (1ll << 31)
        && (vh_g_conv16_force_wm == 8 || (int64_t)(p.M / 256) * (p.N / (p.N % 160 == 0 ? 160 : 128)) >= 256)) {
        const int64_t wgs = (int64_t)(p.M / 256) * (p.N / (p.N % 160 == 0 ? 160 : 128));
        const bool w32 = (p.Wd % 32 == 0) && (p.H % 8 == 0), w16 = (p.Wd % 16 == 0) && (p.H % 16 == 0);      // (round 4: the 16x16-patch form no longer spills and beats the 256-pixel k_conv16 from one workgroup per CU on: 0.122 vs 0.137 ms at 640 -> 640, 16 x 16, B = 64)
        if (w32) return 1;
        if (w16) return 2;
    }
    // 256-pixel tiles (8 waves, two workgroups per CU) once they give every CU a workgroup
    const int64_t big_wgs = (int64_t)((p.M + 255) / 256) * ((p.N + 159) / 160) * nz;
    return (vh_g_conv16_force_wm ? vh_g_conv16_force_wm == 4 : big_wgs >= 256) ? 3 : 0;
}
// timing family of a launch: k_conv16h<5,32> (the decoder's dominant symbol) alone in VH_FAM_CONV16H
static int conv16_family(const Conv16P& p, int nz) { return (pick_conv16(p, nz) == 1 && p.N % 160 == 0) ? VH_FAM_CONV16H : VH_FAM_CONV16_SMALL; }
static int dispatch_conv16(Conv16P& p, int nz, hipStream_t s) {
    const int pick = pick_conv16(p, nz);
    if (p.gn_table) {                                             // the GroupNorm-fused form exists in the halo-patch kernel only (callers ask varhip_conv16_gn_fusable first)
        if (pick == 1) return p.N % 160 == 0 ? launch_conv16h<5, 32, true>(p, s) : launch_conv16h<4, 32, true>(p, s);
        if (pick == 2) return p.N % 160 == 0 ? launch_conv16h<5, 16, true>(p, s) : launch_conv16h<4, 16, true>(p, s);
        return VARHIP_EINVAL;
    }
    if (pick == 1) return p.N % 160 == 0 ? launch_conv16h<5, 32, false>(p, s) : launch_conv16h<4, 32, false>(p, s);
    if (pick == 2) return p.N % 160 == 0 ? launch_conv16h<5, 16, false>(p, s) : launch_conv16h<4, 16, false>(p, s);
    const bool big = pick == 3;
    if (p.N % 160 == 0) return big ? launch_conv16<5, 3, 4, 2>(p, nz, s) : launch_conv16<5, 4, 2>(p, nz, s);
    if (p.N % 128 == 0) return big ? launch_conv16<4, 3, 4, 2>(p, nz, s) : launch_conv16<4, 4, 2>(p, nz, s);
    if (p.N % 64 == 0) return launch_conv16<2, 4, 2>(p, nz, s);
    return launch_conv16<1, 4, 2>(p, nz, s);
}

static int conv16_checks(const void* in, const void* w, const float* bias, const void* out, int B, int H, int W, int Cin, int Cout) {
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || (Cin & 31) || !bias) return VARHIP_EINVAL;
    if ((int64_t)B * H * W >= (1ll << 31)) return VARHIP_EINVAL;
    if ((((uintptr_t)in | (uintptr_t)w | (uintptr_t)out) & 15)) return VARHIP_EINVAL;
    return 0;
}

// out = conv3x3(in) + bias (+ resid): in [B][H][W][Cin] fp16, w [Cout][3][3][Cin] fp16, bias fp32, resid / out [B][H][W][Cout] fp16
// out_mode 1 / 2: the decoder's last conv — fp32 NCHW, clamped to [-1, 1] (2) or de-normalised to [0, 1] (1); gn_part as in gemm.hip
extern "C" int VH16_FN(conv3x3_nhwc)(const void* in, const void* w, const float* bias, const void* resid, void* out, double* gn_part,
                                       int B, int H, int W, int Cin, int Cout, int out_mode, varhip_stream_t stream) {
    int rc = conv16_checks(in, w, bias, out, B, H, W, Cin, Cout);
    if (rc) return rc;
    if (out_mode < 0 || out_mode > 2 || (out_mode != 0 && (resid || gn_part))) return VARHIP_EINVAL;
    if (gn_part && (!varhip_conv_gn_blocks(H, W, Cout, 0) || (Cout & 3))) return VARHIP_EINVAL;
    if ((int64_t)Cout * 9 * Cin * 2 >= (1ll << 32)) return VARHIP_EINVAL;
    {   const int64_t hw = (int64_t)H * W, sample = hw * Cin;
        if (((255 / hw + 2) * sample + (int64_t)(W + 1) * Cin) * 2 >= (1ll << 31)) return VARHIP_EINVAL; }
    Conv16P p{};
    p.in = (const vh_e16*)in; p.w = (const vh_e16*)w; p.bias = bias; p.out = out; p.resid = (const vh_e16*)resid; p.gn_part = gn_part;
    p.M = B * H * W; p.N = Cout; p.K = 9 * Cin; p.H = H; p.Wd = W; p.Cin = Cin; p.phase = 0; p.out_mode = out_mode; p.sW = 0;
    const double npix = (double)B * H * W;
    VhScope scope(conv16_family(p, 1), (hipStream_t)stream, 2.0 * npix * Cout * 9.0 * Cin,
                  2.0 * (npix * Cin + npix * Cout * (resid ? 2.0 : 1.0) + 9.0 * Cin * Cout));
    return dispatch_conv16(p, 1, (hipStream_t)stream);
}

// out = conv3x3(SiLU?(GroupNorm(in))) + bias (+ resid) in ONE launch: `in` is the RAW map, stats [B][G][2] = (mean, rstd) as varhip_gn_stats_* /
// varhip_gn_stats_part_f32 leave them.  Bit-identical to varhip_gn_apply_* followed by varhip_conv3x3_nhwc_* (out_mode 0).  Only shapes the halo-patch
// kernel takes (varhip_conv16_gn_fusable(...) != 0); anything else returns VARHIP_EINVAL and the caller runs the two launches.
static int conv16_gn_fusable(int B, int H, int W, int Cin, int Cout) {
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || (Cin & 31)) return 0;
    Conv16P p{};
    p.M = B * H * W; p.N = Cout; p.K = 9 * Cin; p.H = H; p.Wd = W; p.Cin = Cin;
    const int pick = pick_conv16(p, 1);
    if (pick != 1 && pick != 2) return 0;
    const size_t lds = pick == 1 ? (Cout % 160 == 0 ? conv16h_lds<5, 32>(Cin) : conv16h_lds<4, 32>(Cin)) : (Cout % 160 == 0 ? conv16h_lds<5, 16>(Cin) : conv16h_lds<4, 16>(Cin));
    return lds <= 80 * 1024 ? 1 : 0;
}
#ifndef VH_BF16
extern "C" int varhip_conv16_gn_fusable(int B, int H, int W, int Cin, int Cout) { return conv16_gn_fusable(B, H, W, Cin, Cout); }
#endif
extern "C" int VH16_FN(gnconv3x3_nhwc)(const void* in, const float* table, int silu,
                                         const void* w, const float* bias, const void* resid, void* out, double* gn_part,
                                         int B, int H, int W, int Cin, int Cout, varhip_stream_t stream) {
    int rc = conv16_checks(in, w, bias, out, B, H, W, Cin, Cout);
    if (rc) return rc;
    if (!table || ((uintptr_t)table & 15) || !conv16_gn_fusable(B, H, W, Cin, Cout)) return VARHIP_EINVAL;
    if (gn_part && (!varhip_conv_gn_blocks(H, W, Cout, 0) || (Cout & 3))) return VARHIP_EINVAL;
    if ((int64_t)Cout * 9 * Cin * 2 >= (1ll << 32)) return VARHIP_EINVAL;
    Conv16P p{};
    p.in = (const vh_e16*)in; p.w = (const vh_e16*)w; p.bias = bias; p.out = out; p.resid = (const vh_e16*)resid; p.gn_part = gn_part;
    p.M = B * H * W; p.N = Cout; p.K = 9 * Cin; p.H = H; p.Wd = W; p.Cin = Cin; p.phase = 0; p.out_mode = 0; p.sW = 0;
    p.gn_table = table; p.gn_silu = silu ? 1 : 0;
    const double npix = (double)B * H * W;
    VhScope scope(conv16_family(p, 1), (hipStream_t)stream, 2.0 * npix * Cout * 9.0 * Cin,
                  2.0 * (npix * Cin + npix * Cout * (resid ? 2.0 : 1.0) + 9.0 * Cin * Cout));
    return dispatch_conv16(p, 1, (hipStream_t)stream);
}

// nearest-2x upsample + conv3x3 as four 2x2 phase convolutions on the low-resolution map: in [B][H/2][W/2][Cin] fp16,
// w_phase [4][Cout][2][2][Cin] fp16 (varhip_upconv_pack_f32, then rounded to fp16), out [B][H][W][Cout] fp16
extern "C" int VH16_FN(upconv_phase)(const void* in, const void* w_phase, const float* bias, void* out, double* gn_part,
                                       int B, int H, int W, int Cin, int Cout, varhip_stream_t stream) {
    int rc = conv16_checks(in, w_phase, bias, out, B, H, W, Cin, Cout);
    if (rc) return rc;
    if ((H & 1) || (W & 1) || (Cout & 3)) return VARHIP_EINVAL;
    if (gn_part && !varhip_conv_gn_blocks(H, W, Cout, 1)) return VARHIP_EINVAL;
    {   const int64_t hw = (int64_t)(H / 2) * (W / 2), sample = hw * Cin;
        if (((255 / hw + 2) * sample + (int64_t)(W / 2 + 1) * Cin) * 2 >= (1ll << 31)) return VARHIP_EINVAL; }
    Conv16P p{};
    p.in = (const vh_e16*)in; p.w = (const vh_e16*)w_phase; p.bias = bias; p.out = out; p.resid = nullptr; p.gn_part = gn_part;
    p.M = B * (H / 2) * (W / 2); p.N = Cout; p.K = 4 * Cin; p.H = H / 2; p.Wd = W / 2; p.Cin = Cin; p.phase = 1; p.out_mode = 0;
    p.sW = (int64_t)Cout * 4 * Cin;
    const double npix = (double)B * H * W;
    VhScope scope(conv16_family(p, 4), (hipStream_t)stream, 2.0 * npix * Cout * 4.0 * Cin,
                  2.0 * (npix * Cin / 4.0 + npix * Cout + 16.0 * Cin * Cout));
    return dispatch_conv16(p, 4, (hipStream_t)stream);
}


// ---- k_gnconv16o: the decoder's tail  norm_out -> SiLU -> conv_out  (basic_vae.py:224-226: GroupNorm(32) + swish + conv3x3 to 3 channels,
// then the caller's clamp / (x + 1) / 2: vqvae.py:63, var.py:190) in ONE pass over the 160-channel map.
// As two launches this was the decoder's worst pair: the apply pass reads and writes the 256x256x160 map (2.7 GB at B = 64: HBM-bound), and
// the 3-channel convolution on the 128-pixel kernel re-reads every input pixel once per tap through L2 (12 GB into LDS for 0.04 TFLOP).
// Here a workgroup of 4 waves owns an 8 x 32 patch of one image.  Per 32-channel chunk its (8+2) x (32+2) halo patch goes
// global -> registers -> (x * sc + sh, SiLU, one rounding to the 16-bit type: k_gn16_apply's arithmetic, operation for operation) -> LDS,
// out-of-image pixels as zeros (the convolution pads the NORMALISED map), and the nine taps read it at nine row offsets; the next chunk's
// loads are in flight while the taps run.  K order = chunk outermost, then tap, one 16x16x32 step each, weights as the A operand: the MFMA
// sequence of k_conv16 — the fused result is bit-identical to gn_apply followed by conv3x3_nhwc (tests).  All weights (Cout x 9 x Cin, 8.6 KB) stay in LDS.
struct GnConvOP {
    const vh_e16* in; const float* stats; const float* gamma; const float* beta; const vh_e16* w; const float* bias; float* out;
    int H, Wd, Cin, N, G, out_mode;
};
__global__ void __launch_bounds__(256) k_gnconv16o(GnConvOP p) {
    constexpr int PH = 8, PW = 32, P = PW + 2, PROWS = (PH + 2) * P, ROWB = 64, PATCH = PROWS * ROWB, NPC = (PROWS * 4 + 255) / 256;      // (2 x 21.25 KB of patch + weights + table = 52.2 KB at 160 -> 3: three workgroups per CU)
    extern __shared__ __attribute__((aligned(16))) char smo[];
    char* const sW = smo + 2 * PATCH;                                   // [N][9 * Cin] halves
    float* const sSc = reinterpret_cast<float*>(sW + ((p.N * 9 * p.Cin * 2 + 15) & ~15));      // [Cin] scale, then [Cin] shift
    float* const sSh = sSc + p.Cin;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tX = p.Wd / PW, tY = p.H / PH, tps = tX * tY;
    const int b = blockIdx.x / tps, trem = blockIdx.x - b * tps, tyi = trem / tX, ty0 = tyi * PH, tx0 = (trem - tyi * tX) * PW;
    const int hw = p.H * p.Wd, K = 9 * p.Cin, nch = p.Cin / 32;
    // prologue: weights and the sample's (scale, shift) per channel to LDS
    for (int e = tid; e < p.N * K / 8; e += 256) *(h8*)(sW + e * 16) = *(const h8*)(p.w + (int64_t)e * 8);
    for (int c = tid; c < p.Cin; c += 256) {
        const float* st = p.stats + ((int64_t)b * p.G + c / (p.Cin / p.G)) * 2;
        const float sc = st[1] * p.gamma[c];
        sSc[c] = sc; sSh[c] = p.beta[c] - st[0] * sc;
    }
    // this thread's pieces of a patch chunk: piece e = tid + 256 k -> patch row e >> 2, 8-channel slot e & 3 (= tid & 3 for every k)
    const int slot = tid & 3;
    uint32_t goff[NPC]; int loff[NPC]; uint32_t okmask = 0;
    const vh_e16* const src = p.in + (int64_t)b * hw * p.Cin + slot * 8;
#pragma unroll
    for (int k = 0; k < NPC; ++k) {
        const int pr = (tid + 256 * k) >> 2, py = pr / P, px = pr - py * P, y = ty0 - 1 + py, x = tx0 - 1 + px;
        const bool in_patch = pr < PROWS, ok = in_patch && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.Wd;
        goff[k] = ok ? (uint32_t)((y * p.Wd + x) * p.Cin) : 0u;
        loff[k] = in_patch ? pr * ROWB + ((slot ^ (((pr >> 2) & 1) << 1)) << 4) : -1;
        okmask |= (ok ? 1u : 0u) << k;
    }
    // two chunks in flight: chunk c + 2 is requested before the taps of chunk c run and parked behind the taps of chunk c + 1 (one
    // chunk ahead left the loads ~600 cycles of MFMA to hide an HBM round trip behind)
    h8 rawA[NPC], rawB[NPC];
    auto fetch = [&](int c, h8* raw) {
#pragma unroll
        for (int k = 0; k < NPC; ++k) raw[k] = *(const h8*)(src + goff[k] + c * 32);          // (pieces past the patch / outside the image: offset 0, never parked / parked as zeros)
    };
    auto park = [&](int c, const h8* raw) {                             // normalise + SiLU + round, as k_gn16_apply; zeros outside the image
        const f32x4 s0 = *(const f32x4*)(sSc + c * 32 + slot * 8), s1 = *(const f32x4*)(sSc + c * 32 + slot * 8 + 4);
        const f32x4 h0 = *(const f32x4*)(sSh + c * 32 + slot * 8), h1 = *(const f32x4*)(sSh + c * 32 + slot * 8 + 4);
        char* const dst = smo + (c & 1) * PATCH;
#pragma unroll
        for (int k = 0; k < NPC; ++k) {
            if (loff[k] < 0) continue;
            float z[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float y = __builtin_fmaf((float)raw[k][e], e < 4 ? s0[e & 3] : s1[e & 3], e < 4 ? h0[e & 3] : h1[e & 3]);
                z[e] = y * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * y));
            }
            // the product rounds to fp32 BEFORE the conversion, as in k_gn16_apply (hipcc otherwise merges the two into v_fma_mixlo_f16, which
            // rounds once: other bits in 1 of 10^4 elements)
            asm volatile("" : "+v"(z[0]), "+v"(z[1]), "+v"(z[2]), "+v"(z[3]), "+v"(z[4]), "+v"(z[5]), "+v"(z[6]), "+v"(z[7]));
            h8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (vh_e16)z[e];
            if (!((okmask >> k) & 1u)) o = (h8)(vh_e16)0.0f;
            *(h8*)(dst + loff[k]) = o;
        }
    };
    fetch(0, rawA);
    if (nch > 1) fetch(1, rawB);
    __syncthreads();                                                    // scale / shift table (and the weights) are in LDS
    park(0, rawA);
    __syncthreads();

    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int r16 = lane & 15, kq = lane >> 4;
    const int rb = wave * 2 * P + r16;                                  // patch row of this lane's pixel of fragment 0 at tap (0, 0): the wave owns output rows 2 wave, 2 wave + 1
    const char* const wl = sW + (size_t)(r16 < p.N ? r16 : p.N - 1) * K * 2 + kq * 16;      // (rows past Cout repeat the last one: those output channels are never stored)
    auto taps = [&](int c) {
        const char* const pa = smo + (c & 1) * PATCH;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int ky = t / 3, kx = t - ky * 3;
            const h8 bn = *(const h8*)(wl + (size_t)(t * p.Cin + c * 32) * 2);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int R = rb + ((i >> 1) + ky) * P + (i & 1) * 16 + kx;
                const h8 am = *(const h8*)(pa + R * ROWB + ((kq << 4) ^ (((R >> 2) & 1) << 5)));
                acc[i] = VH16_MFMA_16x16x32(bn, am, acc[i]);
            }
        }
    };
    // chunk c (even: rawA held it, rawB holds c + 1; odd: the other way round).  The buffer park() writes was read last in chunk c - 1,
    // which every wave left at the barrier that ends it.
    for (int c = 0; c < nch; c += 2) {
        if (c + 2 < nch) fetch(c + 2, rawA);
        taps(c);
        if (c + 1 < nch) park(c + 1, rawB);
        __syncthreads();
        if (c + 1 >= nch) break;
        if (c + 3 < nch) fetch(c + 3, rawB);
        taps(c + 1);
        if (c + 2 < nch) park(c + 2, rawA);
        __syncthreads();
    }
    // epilogue: acc[i][e] = C[pixel (row 2 wave + (i >> 1), column 16 (i & 1) + r16)][channel 4 kq + e]; fp32 NCHW, clamp (+ de-normalise)
    if (kq * 4 < p.N) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int y = ty0 + wave * 2 + (i >> 1), x = tx0 + (i & 1) * 16 + r16;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int n = kq * 4 + e;
                if (n >= p.N) break;
                const float v = acc[i][e] + p.bias[n];
                const float cl = vm_min(vm_max(v, -1.0f), 1.0f);
                p.out[((int64_t)b * p.N + n) * hw + y * p.Wd + x] = p.out_mode == 1 ? (cl + 1.0f) * 0.5f : cl;
            }
        }
    }
}

// out = clamp(conv3x3(SiLU(GroupNorm(x))) + bias) in fp32 NCHW (out_mode 2), de-normalised to [0, 1] (out_mode 1): x [B][H][W][Cin] 16-bit
// channels-last, stats [B][G][2] = (mean, rstd) as varhip_gn_stats_* / varhip_gn_stats_part_f32 leave them, w [Cout][3][3][Cin] 16-bit.
// Takes maps that tile into 8 x 32 patches with Cin % 32 == 0 and Cout <= 16; anything else returns VARHIP_EINVAL (the caller then runs
// varhip_gn_apply_* + varhip_conv3x3_nhwc_*, which this call equals bit for bit).
extern "C" int VH16_FN(gn_silu_conv_out)(const void* x, const float* stats, const float* gamma, const float* beta, const void* w, const float* bias,
                                           float* out, int B, int H, int W, int Cin, int Cout, int G, int out_mode, varhip_stream_t stream) {
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || G <= 0 || !x || !stats || !gamma || !beta || !w || !bias || !out) return VARHIP_EINVAL;
    if ((H % 8) || (W % 32) || (Cin % 32) || (Cin % G) || Cout > 16 || (out_mode != 1 && out_mode != 2)) return VARHIP_EINVAL;
    if ((((uintptr_t)x | (uintptr_t)w) & 15) || (int64_t)H * W * Cin * 2 >= (1ll << 31) || (int64_t)B * (H / 8) * (W / 32) >= (1ll << 31)) return VARHIP_EINVAL;
    constexpr int PATCH = 10 * 34 * 64;
    const size_t lds = (size_t)2 * PATCH + (((size_t)Cout * 9 * Cin * 2 + 15) & ~(size_t)15) + (size_t)2 * Cin * 4;
    if (lds > 64 * 1024) return VARHIP_EINVAL;
    GnConvOP p{(const vh_e16*)x, stats, gamma, beta, (const vh_e16*)w, bias, out, H, W, Cin, Cout, G, out_mode};
    static bool attr_done = false;
    if (!attr_done) { (void)hipFuncSetAttribute((const void*)k_gnconv16o, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024); attr_done = true; }
    const double npix = (double)B * H * W;
    VhScope scope(VH_FAM_CONV16_SMALL, (hipStream_t)stream, 2.0 * npix * Cout * 9.0 * Cin, 2.0 * npix * Cin + 4.0 * npix * Cout + 18.0 * Cin * Cout);
    hipLaunchKernelGGL(k_gnconv16o, dim3(B * (H / 8) * (W / 32)), dim3(256), lds, (hipStream_t)stream, p);
    return vh_launch_status();
}

}  // namespace VH16_NS
