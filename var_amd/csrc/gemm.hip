// gemm.hip — fp32 MFMA GEMM (out = epi(A . W^T + bias)) and implicit-GEMM 3x3 convolution for gfx950.
//
// Arithmetic contract: each output element is ONE k-ascending fp32 fma chain from 0 — which is exactly what a sequence of
// fp32 MFMAs (v_mfma_f32_16x16x4_f32 or 32x32x2) into one accumulator computes (cdna_hip_programming.md §3 "FP32-input
// MFMA").  So no split-K and no reordering of k: the result is bit-identical to oracle/var_oracle.c.
//
// Two kernels: k_dma_gemm (everything on the hot path: transformer linears, the fused q/k/v GEMM, all 3x3 convolutions of
// the VQVAE as implicit GEMMs) and k_gemm_any (shapes the DMA kernel cannot take).  Both operands are K-contiguous ("NT").
#include "common.h"

struct GemmP {
    const float* A; const float* W; const float* bias; float* out; const float* resid; const float* gamma;
    int64_t lda, ldw, ldo, ldr, ldg, sA, sW, sO;
    int M, N, K, epi, rows_per_group, bias_per_row;
    int H, Wd, Cin, up2, out_mode, Hi, Wi;     // convolution only
    int tilesM, tilesN;
    int evec;                                   // epilogue may use 16-byte accesses (N, leading dims and pointers allow it)
    // epi == 3 (fused q/k/v epilogue, varhip_gemm_qkv_f32): N = 3C, head_dim 64
    const float* q_smul; float* q_out; float* q_kc; float* q_vc; float q_plain; int q_l2, q_l, q_pos0, q_Lmax;
    // convolution: optional per-block per-channel (sum, sum of squares) of the result, [B][blocks per sample][Cout][2] doubles, for
    // the GroupNorm that follows (saves its statistics pass over the tensor)
    double* gn_part;
};

// ------------------------------------------------------------------------------------------------------------------
// k_gemm_any — fallback for operands the DMA kernel cannot take (K % 32 != 0, odd leading dimensions, unaligned pointers):
// the tiny VAE-attention products of small test configurations.  Same arithmetic contract (one k-ascending fma chain per
// output: MFMA 32x32x2 walks k = 2s + (lane >> 5)), operands fetched straight from global memory with guarded scalar loads.
// 4 waves = 2 x 2 tiles of 32 x 32.
__global__ void __launch_bounds__(256) k_gemm_any(GemmP p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
    const int tm = blockIdx.x / p.tilesN, tn = blockIdx.x - tm * p.tilesN, bz = blockIdx.z;
    const int m0 = tm * 64 + (wave >> 1) * 32, n0 = tn * 64 + (wave & 1) * 32;
    const int ma = m0 + r, nb = n0 + r;
    const float* a = p.A + (int64_t)bz * p.sA + (int64_t)(ma < p.M ? ma : 0) * p.lda;
    const float* w = p.W + (int64_t)bz * p.sW + (int64_t)(nb < p.N ? nb : 0) * p.ldw;
    float* Ob = p.out + (int64_t)bz * p.sO;
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    for (int k0 = 0; k0 < p.K; k0 += 2) {
        const int k = k0 + h;
        const float av = (ma < p.M && k < p.K) ? a[k] : 0.f, wv = (nb < p.N && k < p.K) ? w[k] : 0.f;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, wv, acc, 0, 0, 0);
    }
    const int n = n0 + r;                                          // C/D layout: col = lane & 31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
    if (n >= p.N) return;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int m = m0 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (m >= p.M) continue;
        float v = acc[e];
        if (p.bias) v = v + (p.bias_per_row ? p.bias[m] : p.bias[n]);
        if (p.epi == VARHIP_EPI_GELU) v = vm_gelu_tanh(v);
        else if (p.epi == VARHIP_EPI_RESID) {
            if (p.gamma) v = v * p.gamma[(int64_t)(m / p.rows_per_group) * p.ldg + n];
            v = p.resid[(int64_t)m * p.ldr + n] + v;
        }
        Ob[(int64_t)m * p.ldo + n] = v;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// k_dma_gemm — the dense GEMM of the transformer (K % 32 == 0, 16-byte aligned operands): same arithmetic contract, leaner
// data path.  Both operand tiles go global -> LDS with the LDS-DMA instruction (global_load_lds, 16 B per lane): no staging
// VGPRs, no ds_write.  One instruction moves 8 rows x 128 B; a lane cannot choose its LDS slot (slot = lane), so the
// bank-conflict swizzle is applied on the SOURCE side: slot c of row r receives 16-byte chunk c ^ (r & 7), and the reader
// of k-step s looks in slot s ^ (r & 7).
// MFMA 16x16x4 (4 k per instruction, k = lane >> 4 -> one ds_read_b32 per operand tile per step, natural k order, the
// same k-ascending fma chain per output as the 32x32x2 form).  The WEIGHT tile is the A operand, so D[row = n][col = m]:
// a lane ends up with 4 consecutive n of one m and the epilogue (bias / GELU / gamma / residual / q-k-v prep) runs on float4s
// straight from the accumulators, no LDS transpose.
// 2x2 waves; a wave owns (TMW*16) x (TNW*16) outputs.  Two LDS stages, one barrier per K tile.
__device__ __attribute__((aligned(128))) float g_zero_row[32];      // zero padding source of the nearest-2x GATHER mode only (never written)

template <int N> __device__ __forceinline__ void vh_waitcnt_barrier() { asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory"); }
template <int MAXA, int PER> __device__ __forceinline__ void vh_wait_dma_and_barrier(int ahead) {      // `ahead` is wave-uniform
    if constexpr (MAXA == 0) vh_waitcnt_barrier<0>();
    else { if (ahead >= MAXA) vh_waitcnt_barrier<MAXA * PER>(); else vh_wait_dma_and_barrier<MAXA - 1, PER>(ahead); }
}

// One LDS-DMA request in the "scalar base + 32-bit vector offset" form: 16 bytes per lane from base + voff to LDS address `lds`
// (+ 16 * lane).  Written as inline assembly because the compiler hoists the zero-extension of the offset out of the K loop and
// then only sees a 64-bit vector address (v_lshl_add_u64 per request).  The waitcnt pass does not see these requests: every
// barrier that publishes a tile spells out its own s_waitcnt vmcnt (vh_waitcnt_barrier).  M0 (the LDS destination) is written
// inside the statement that uses it and named in the clobber list, so the compiler never assumes a value of its own survives.
__device__ __forceinline__ void vh_dma16(const void* base, uint32_t voff, uint32_t lds) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(base), "s"(lds) : "memory", "m0");
}

// The same through a buffer descriptor: 16 bytes per lane from rsrc.base + soff + voff.  A request whose voff lies at or beyond
// rsrc.num_records is out of range for the hardware bounds check and delivers zeros to its LDS slot — the zero padding of the
// convolution costs one OR into the offset instead of a 64-bit address select.
__device__ __forceinline__ void vh_dma16_buf(__amdgpu_buffer_rsrc_t rsrc, uint32_t voff, uint32_t soff, uint32_t lds) {
    // (readfirstlane: a value the compiler computed on the vector ALU although it is wave-uniform must still reach an SGPR operand)
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds"
                 : : "v"(voff), "s"(rsrc), "s"(__builtin_amdgcn_readfirstlane(soff)), "s"(__builtin_amdgcn_readfirstlane(lds)) : "memory", "m0");
}

// The same with a full 64-bit per-lane source address (the nearest-2x GATHER mode, whose source is not "base + offset").  Every LDS-DMA
// of this file is one of these three statements: each writes M0 itself and declares it clobbered, and the compiler's own
// M0-tracking builtin (__builtin_amdgcn_global_load_lds) is not used next to them.
__device__ __forceinline__ void vh_dma16_ptr(const void* src, uint32_t lds) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(lds) : "memory", "m0");
}

// GATHER: the nearest-2x gather mode of the convolution (up2 == 1; tests and the oracle comparison only — the decoder runs the
// phase form), kept out of the common kernels so that their tap address stays two adds and a select
template <int TMW, int TNW, bool CONV, int NST = 2, bool GATHER = false>
__global__ void __launch_bounds__(256) k_dma_gemm(GemmP p) {
    constexpr int BK = 32, BM = TMW * 32, BN = TNW * 32, STAGE = (BM + BN) * BK;
    constexpr int NIA = BM / 32, NIB = BN / 32;                   // DMA instructions per wave and K tile
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // the wave index is read into a scalar register: the LDS-DMA destination (M0) and all per-wave tile offsets then stay on the
    // scalar unit instead of going VGPR -> v_readfirstlane -> M0 in front of every DMA instruction
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
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
    const int m0 = tm_ * BM, n0 = tn_ * BN;
    const int bz = blockIdx.z;
    const float* Ab = p.A + (int64_t)bz * p.sA;
    const float* Wb = p.W + (int64_t)bz * p.sW;
    float* Ob = p.out + (int64_t)bz * p.sO;

    // DMA roles: wave w stages rows [w*BM/4, (w+1)*BM/4) of the activation tile and [w*BN/4, ...) of the weight tile.
    // Rows past M / N are clamped to the last valid row: they only feed outputs that are never stored.
    const int drow = lane >> 3, dslot = lane & 7;
    // GEMM-mode activations and all weights are addressed as "uniform base (SGPRs) + per-lane 32-bit byte offset": the offsets are
    // loop constants and advancing K is scalar arithmetic, so a DMA request costs no vector-ALU instruction (the 64-bit per-lane
    // address adds it replaces sat in the MFMA stream: ~20 of them per K tile).  The host checks that the offsets fit 32 bits.
    const float* asrc[NIA]; uint32_t aoff[NIA], boff[NIB];
    // CONV: a_y/a_x = the row's pixel in source coordinates before the tap offset (2y, 2x for the stride-2 mode); asrc = address of
    // that pixel's channel 0 (+ swizzle), so a tap is one scalar offset away: ((dy*Wi + dx)*Cin + ci0).  Only the nearest-2x
    // gather mode (up2 == 1), whose source index is not linear in the tap, recomputes the full address (a_b kept for it).
    int a_b[NIA], a_y[NIA], a_x[NIA];
    // CONV without the gather: the input is read through a buffer descriptor whose window starts one row + one pixel before the
    // first sample the tile touches (so every tap offset is a non-negative scalar) and is < 2 GB long (host-checked); aoff = the
    // pixel's byte offset in that window, abad bit t = tap t of this pixel is zero padding.  A padded tap ORs bit 31 into the
    // offset, which puts the request out of range of the window: the bounds check writes zeros to the LDS slot.
    uint32_t abad[NIA];
    const float* rs_base = Ab; int64_t rs_bytes = 0;
    const int cv_pre = (CONV && p.up2 != 3) ? 1 : 0;              // taps reach one row / column back (not in the stride-2 mode)
    const int cv_ntap = CONV ? p.K / p.Cin : 1, cv_hlim = CONV ? (p.up2 == 3 ? p.Hi : p.H) : 0, cv_wlim = CONV ? (p.up2 == 3 ? p.Wi : p.Wd) : 0;
    int cv_b0 = 0;
    if constexpr (CONV && !GATHER) {
        const int hw = p.H * p.Wd, mlast = (m0 + BM - 1 < p.M ? m0 + BM - 1 : p.M - 1), mfirst = m0 < p.M ? m0 : p.M - 1;
        cv_b0 = mfirst / hw;
        const int64_t sample = (int64_t)p.Hi * p.Wi * p.Cin, shift = (int64_t)cv_pre * (p.Wi + 1) * p.Cin;
        rs_bytes = ((int64_t)(mlast / hw - cv_b0 + 1) * sample + shift) * 4;
        rs_base = Ab + (int64_t)cv_b0 * sample - shift;
    }
    const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc((void*)rs_base, 0, (int)rs_bytes, 0x00020000);
    // pixel coordinates of the lane's NIA rows (8 apart): one division pair for the first, the others by stepping — the prologue's
    // integer divisions and tap tests are paid by every workgroup while its SIMDs' matrix pipes wait
    const bool cv_step = CONV && p.Wd >= 8 && m0 + BM <= p.M;
#pragma unroll
    for (int i = 0; i < NIA; ++i) {
        int m = m0 + wave * (BM / 4) + i * 8 + drow; m = m < p.M ? m : p.M - 1;
        if (CONV) {
            const int hw = p.H * p.Wd;
            if (i == 0 || !cv_step) {
                a_b[i] = m / hw; const int rem2 = m - a_b[i] * hw; a_y[i] = rem2 / p.Wd; a_x[i] = rem2 - a_y[i] * p.Wd;
            } else {
                a_b[i] = a_b[i - 1]; a_y[i] = a_y[i - 1]; a_x[i] = a_x[i - 1] + 8;
                if (a_x[i] >= p.Wd) { a_x[i] -= p.Wd; if (++a_y[i] == p.H) { a_y[i] = 0; ++a_b[i]; } }
            }
        } else {
            aoff[i] = (uint32_t)(((int64_t)m * p.lda + ((dslot ^ drow) << 2)) * 4);
        }
    }
    if constexpr (CONV) {
#pragma unroll
        for (int i = 0; i < NIA; ++i) {
            if (p.up2 == 3) { a_y[i] *= 2; a_x[i] *= 2; }
            asrc[i] = Ab + ((dslot ^ drow) << 2);
            if constexpr (!GATHER) {
                aoff[i] = (uint32_t)(((((int64_t)(a_b[i] - cv_b0) * p.Hi + a_y[i]) * p.Wi + a_x[i]) * p.Cin + ((dslot ^ drow) << 2)) * 4);
                // tap t = 3*ky + kx (2*a + b in the phase mode) pads iff its row offset or its column offset leaves the image
                const int dy0 = p.up2 == 2 ? (bz >> 1) - 1 : -cv_pre, dx0 = p.up2 == 2 ? (bz & 1) - 1 : -cv_pre;
                const bool r0 = (unsigned)(a_y[i] + dy0) >= (unsigned)cv_hlim, r1 = (unsigned)(a_y[i] + dy0 + 1) >= (unsigned)cv_hlim,
                           r2 = (unsigned)(a_y[i] + dy0 + 2) >= (unsigned)cv_hlim;
                const bool c0 = (unsigned)(a_x[i] + dx0) >= (unsigned)cv_wlim, c1 = (unsigned)(a_x[i] + dx0 + 1) >= (unsigned)cv_wlim,
                           c2 = (unsigned)(a_x[i] + dx0 + 2) >= (unsigned)cv_wlim;
                abad[i] = p.up2 == 2 ? ((r0 ? 0x3u : 0u) | (r1 ? 0xCu : 0u) | (c0 ? 0x5u : 0u) | (c1 ? 0xAu : 0u))
                                     : ((r0 ? 0x007u : 0u) | (r1 ? 0x038u : 0u) | (r2 ? 0x1C0u : 0u) | (c0 ? 0x049u : 0u) | (c1 ? 0x092u : 0u) | (c2 ? 0x124u : 0u));
            }
        }
    }
    const float* zsrc = g_zero_row + ((dslot ^ drow) << 2);
#pragma unroll
    for (int i = 0; i < NIB; ++i) {
        int n = n0 + wave * (BN / 4) + i * 8 + drow; n = n < p.N ? n : p.N - 1;
        boff[i] = (uint32_t)(((int64_t)n * p.ldw + ((dslot ^ drow) << 2)) * 4);
    }
    int cv_tap = 0, cv_cc = 0;                                    // CONV: tap / channel chunk of the next K tile to be requested
    auto dma_tile = [&](int kt, int st) {                         // K tiles must be requested in order 0, 1, 2, ...
        float* sA = smem + st * STAGE + wave * (BM / 4) * BK;
        float* sB = smem + st * STAGE + BM * BK + wave * (BN / 4) * BK;
        int dy = 0, dx = 0, ci0 = 0, woff = kt * BK, tap = 0;
        uint32_t soff = 0;
        if (CONV) {
            // Summation order of the convolutions: 32-channel chunks outermost, then the taps, then the channels of the chunk
            // (K tile kt = chunk kt / ntap, tap kt % ntap; the tiles are requested in order, so two counters replace the division).
            // The 9 (or 4) consecutive K tiles of a chunk read the same few cache lines of the input, so the tap re-reads stay
            // inside the XCD's L2: tap-major order moved 6.7 GB per 256x256 launch across the fabric, this order 2.4 GB
            // (algorithmic 1.6 GB; profiles/r01_pmc_traffic.json).  Weights keep the [Cout][tap][Cin] layout: the tile's weights
            // are the 32 floats at tap*Cin + ci0 of every row.
            tap = cv_tap;
            ci0 = cv_cc * BK;
            woff = tap * p.Cin + ci0;
            if (++cv_tap == cv_ntap) { cv_tap = 0; ++cv_cc; }
            if (p.up2 == 2) { dy = (tap >> 1) - 1 + (bz >> 1); dx = (tap & 1) - 1 + (bz & 1); }   // phase (bz>>1, bz&1) of the folded Upsample2x conv
            else { const int ky = tap / 3; dy = ky - (p.up2 == 3 ? 0 : 1); dx = tap - ky * 3 - (p.up2 == 3 ? 0 : 1); }
            soff = (uint32_t)(((dy + cv_pre) * p.Wi + dx + cv_pre) * p.Cin + ci0) * 4u;      // tap offset inside the descriptor window
        }
#pragma unroll
        for (int i = 0; i < NIA; ++i) {
            // up2: 0 plain, 1 nearest-2x gather, 2 phase conv on the low-res map, 3 stride 2 over an input zero-padded at the bottom/right
            if constexpr (CONV && GATHER) {
                const int yy = a_y[i] + dy, xx = a_x[i] + dx;
                const bool ok = (unsigned)yy < (unsigned)cv_hlim && (unsigned)xx < (unsigned)cv_wlim;
                const float* src = ok ? asrc[i] + (((int64_t)a_b[i] * p.Hi + (yy >> 1)) * p.Wi + (xx >> 1)) * p.Cin + ci0 : zsrc;
                vh_dma16_ptr(src, (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)(sA + i * 8 * BK));
            } else if constexpr (CONV) {
                vh_dma16_buf(arsrc, ((abad[i] >> tap) << 31) | aoff[i], soff,
                             (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)(sA + i * 8 * BK));
            } else {
                vh_dma16((const char*)Ab + (size_t)kt * (BK * 4), aoff[i],
                         (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)(sA + i * 8 * BK));
            }
        }
#pragma unroll
        for (int i = 0; i < NIB; ++i)
            vh_dma16((const char*)Wb + (size_t)woff * 4, boff[i], (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)(sB + i * 8 * BK));
    };

    f32x4 acc[TMW][TNW];
#pragma unroll
    for (int i = 0; i < TMW; ++i)
#pragma unroll
        for (int j = 0; j < TNW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int r16 = lane & 15, kq = lane >> 4;
    const int nk = p.K / BK;
    // `mid()` (the request for the next K tile) runs after the first step's MFMAs are issued: the wave comes out of the barrier
    // straight into LDS reads and matrix work, and the DMA address/M0 bookkeeping executes in the shadow of those MFMAs
    // a wave whose whole column range lies past N (the 160 -> 3 output conv has N = 3 in a 32-wide tile, so half the waves) or whose
    // rows lie past M only keeps up the DMA requests and the barriers; its SIMD's matrix time goes to the other workgroups on the CU
    const bool idle_wave = (n0 + wn * TNW * 16 >= p.N) || (m0 + wm * TMW * 16 >= p.M);
    auto compute = [&](int cur, auto&& mid) {
        const float* sA = smem + cur * STAGE + (wm * TMW * 16 + r16) * BK + kq;
        const float* sB = smem + cur * STAGE + BM * BK + (wn * TNW * 16 + r16) * BK + kq;
        if constexpr (TMW * TNW == 1) {
            // one accumulator per wave: the MFMAs form a dependent chain and a step would otherwise wait one LDS round trip;
            // fetch the fragments of all eight steps first
            float a8[8], b8[8];
#pragma unroll
            for (int s = 0; s < 8; ++s) { const int sl = (s ^ (r16 & 7)) << 2; a8[s] = sA[sl]; b8[s] = sB[sl]; }
            mid();
#pragma unroll
            for (int s = 0; s < 8; ++s) acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(b8[s], a8[s], acc[0][0], 0, 0, 0);
            return;
        }
        // operand fragments of step s+1 are read while the MFMAs of step s run (two register sets; the scheduling barrier keeps
        // the compiler from sinking the reads back next to their use): +2..3 % on the 128-wide tiles, neutral on 64x64
        float am[2][TMW], bn[2][TNW];
        {
            const int sl = (r16 & 7) << 2;                        // rows i*16 + r16: (row & 7) == (r16 & 7)
#pragma unroll
            for (int i = 0; i < TMW; ++i) am[0][i] = sA[i * 16 * BK + sl];
#pragma unroll
            for (int j = 0; j < TNW; ++j) bn[0][j] = sB[j * 16 * BK + sl];
        }
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            if (s + 1 < 8) {
                const int sl = ((s + 1) ^ (r16 & 7)) << 2;
#pragma unroll
                for (int i = 0; i < TMW; ++i) am[(s + 1) & 1][i] = sA[i * 16 * BK + sl];
#pragma unroll
                for (int j = 0; j < TNW; ++j) bn[(s + 1) & 1][j] = sB[j * 16 * BK + sl];
            }
#pragma unroll
            for (int i = 0; i < TMW; ++i)
#pragma unroll
                for (int j = 0; j < TNW; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bn[s & 1][j], am[s & 1][i], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (s == 0) { mid(); __builtin_amdgcn_sched_barrier(0); }
        }
    };
    if constexpr (NST == 2) {
        dma_tile(0, 0);
        vh_waitcnt_barrier<0>();                                  // wait for this wave's own DMA (vmcnt), then publish
        if (idle_wave) {                                          // same requests and barriers, no matrix work (kept out of the hot loop)
            for (int kt = 0; kt < nk; ++kt) { if (kt + 1 < nk) dma_tile(kt + 1, (kt & 1) ^ 1); vh_waitcnt_barrier<0>(); }
        } else {
            // two K tiles per trip, so the LDS stage is a compile-time constant in each copy of the tile body: the operand reads
            // then take their stage/tile offsets as instruction immediates instead of one v_add per read
            int kt = 0;
            for (; kt + 1 < nk; kt += 2) {
                compute(0, [&] { dma_tile(kt + 1, 1); });                  // stage 1 was last read before the previous barrier
                vh_waitcnt_barrier<0>();
                compute(1, [&] { if (kt + 2 < nk) dma_tile(kt + 2, 0); });
                vh_waitcnt_barrier<0>();
            }
            if (kt < nk) { compute(0, [] {}); vh_waitcnt_barrier<0>(); }   // odd tile count: the last tile sits in stage 0
        }
    } else {
        // Deep pipeline for launches with fewer workgroups than CUs (small scales): NST-1 tiles in flight, so the K loop of the
        // lone workgroup on a CU runs at the DMA issue rate, not at one memory latency per tile.  LDS-DMA completes in order per
        // wave: waiting until at most `ahead` later tiles are outstanding means tile kt has landed; the barrier then publishes all
        // four waves' parts and doubles as the "stage (kt-1) % NST is free" signal for the next issue.  (A __syncthreads() would
        // drain every DMA.)
        constexpr int PER = NIA + NIB;
        static_assert((NST - 2) * PER <= 63, "vmcnt is a 6-bit counter");
#pragma unroll
        for (int t = 0; t < NST - 1; ++t) if (t < nk) dma_tile(t, t);
        for (int kt = 0; kt < nk; ++kt) {
            const int ahead = (nk - 1 - kt) < (NST - 2) ? (nk - 1 - kt) : (NST - 2);
            vh_wait_dma_and_barrier<NST - 2, PER>(ahead);
            compute(kt % NST, [&] { if (kt + NST - 1 < nk) dma_tile(kt + NST - 1, (kt + NST - 1) % NST); });
        }
    }

    // ---- epilogue: acc[i][j][e] = C[m = tile_m(i) + r16][n = tile_n(j) + 4*kq + e]
    const int nw0 = n0 + wn * TNW * 16;
    if (nw0 >= p.N) return;
    if constexpr (!CONV && TNW == 4) {
        if (p.epi == 3) {
            // fused q/k/v post-processing (SelfAttention.forward up to the cache append, basic_var.py:98-109): the wave's 64 columns
            // are one head of q, k or v.  Sum of squares in the canonical W64 butterfly order of k_qkv_prep with element c = channel =
            // 16j + 4kq + e: offsets 32,16 pair the j tiles (registers), 8,4 pair kq (lanes ^32, ^16), 2,1 pair e (registers).
            const int C = p.N / 3, sect = nw0 / C, head = (nw0 - sect * C) >> 6, Hh = C >> 6;
            f32x4 b4[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) b4[j] = p.bias ? *(const f32x4*)(p.bias + nw0 + j * 16 + kq * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
            const float sm = (p.q_l2 && sect == 0) ? vm_exp(vm_min(p.q_smul[head], 4.605170249938965f)) : 1.0f;
#pragma unroll
            for (int i = 0; i < TMW; ++i) {
                const int m = m0 + (wm * TMW + i) * 16 + r16;
                f32x4 v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = acc[i][j] + b4[j];
                if (p.q_l2 && sect < 2) {
                    const f32x4 a0 = v[0] * v[0] + v[2] * v[2], a1 = v[1] * v[1] + v[3] * v[3];     // offset 32
                    f32x4 b = a0 + a1;                                                               // offset 16
#pragma unroll
                    for (int e = 0; e < 4; ++e) b[e] = b[e] + __shfl_xor(b[e], 32, 64);              // offset 8
#pragma unroll
                    for (int e = 0; e < 4; ++e) b[e] = b[e] + __shfl_xor(b[e], 16, 64);              // offset 4
                    const float t0 = b[0] + b[2], t1 = b[1] + b[3];                                  // offset 2
                    const float den = vm_max(vm_sqrt(t0 + t1), 1e-12f);                              // offset 1
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[j][e] = sect == 0 ? (v[j][e] / den) * sm : v[j][e] / den;
                } else if (!p.q_l2 && sect == 0) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = v[j] * p.q_plain;
                }
                if (m >= p.M) continue;
                float* dst;
                if (sect == 0) dst = p.q_out + (int64_t)m * C + head * 64;
                else {
                    const int bb = m / p.q_l, t = m - bb * p.q_l;
                    dst = (sect == 1 ? p.q_kc : p.q_vc) + (((int64_t)bb * Hh + head) * p.q_Lmax + p.q_pos0 + t) * 64;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) *(f32x4*)(dst + j * 16 + kq * 4) = v[j];
            }
            return;
        }
    }
    // Lean epilogue of full tiles.  Vector-ALU work here is paid at several cycles per instruction (the other workgroups of the CU are
    // in their matrix loops), so a row gets ONE 32-bit byte offset per operand relative to the tile's first row (scalar 64-bit bases),
    // the column tiles are instruction immediates on "scalar base + offset" loads and stores, and the AdaLN gate's row group comes
    // from one scalar division when a tile spans at most two groups.  Same arithmetic, element for element, as the general loop below.
    bool lean_done = false;
    if constexpr (!CONV) {
        if (p.evec && p.bias && !p.bias_per_row && m0 + BM <= p.M && n0 + BN <= p.N && ((p.ldo | p.ldr | p.ldg) >> 22) == 0) {   // (row offsets of a tile fit 32 bits)
            const int col = nw0 + kq * 4;
            const bool res = p.epi == VARHIP_EPI_RESID, gam = res && p.gamma;
            const int rpg = p.rows_per_group, g0 = m0 / rpg;
            uint32_t ooff[TMW], roff[TMW], goff[TMW];
#pragma unroll
            for (int i = 0; i < TMW; ++i) {
                const int dm = (wm * TMW + i) * 16 + r16, m = m0 + dm;
                ooff[i] = ((uint32_t)dm * (uint32_t)p.ldo + (uint32_t)col) * 4u;
                roff[i] = res ? ((uint32_t)dm * (uint32_t)p.ldr + (uint32_t)col) * 4u : 0u;
                const int dg = rpg >= BM ? (m >= (g0 + 1) * rpg ? 1 : 0) : m / rpg - g0;
                goff[i] = gam ? ((uint32_t)dg * (uint32_t)p.ldg + (uint32_t)col) * 4u : 0u;
            }
            char* ob = (char*)(Ob + (int64_t)m0 * p.ldo);
            const char* rb = res ? (const char*)(p.resid + (int64_t)m0 * p.ldr) : nullptr;
            const char* gb = gam ? (const char*)(p.gamma + (int64_t)g0 * p.ldg) : nullptr;
#pragma unroll
            for (int j = 0; j < TNW; ++j) {
                const f32x4 b4 = *(const f32x4*)(p.bias + col + j * 16);
#pragma unroll
                for (int i = 0; i < TMW; ++i) {
                    f32x4 v = acc[i][j] + b4;
                    if (p.epi == VARHIP_EPI_GELU) {
                        const f32x2 lo = vh_gelu_tanh_pair(f32x2{v[0], v[1]}), hi = vh_gelu_tanh_pair(f32x2{v[2], v[3]});
                        v = f32x4{lo[0], lo[1], hi[0], hi[1]};
                    }
                    if (res) {
                        if (gam) v = v * *(const f32x4*)(gb + (uint64_t)goff[i] + j * 64);
                        v = *(const f32x4*)(rb + (uint64_t)roff[i] + j * 64) + v;
                    }
                    *(f32x4*)(ob + (uint64_t)ooff[i] + j * 64) = v;
                }
            }
            return;
        }
    }
    if constexpr (CONV && NST == 2) {
        // convolutions (NHWC output, full tile): the same, plus the scattered rows of the phase mode and the GroupNorm partial sums
        if (p.evec && p.out_mode == 0 && m0 + BM <= p.M && n0 + BN <= p.N) {
            const int col = nw0 + kq * 4, hw = p.H * p.Wd;
            const bool res = p.epi == VARHIP_EPI_RESID, phase = p.up2 == 2;
            int64_t mo0 = m0;                                          // output row of the tile's first pixel
            if (phase) { const int b = m0 / hw, rem = m0 - b * hw, y = rem / p.Wd, x = rem - y * p.Wd;
                         mo0 = ((int64_t)b * (2 * p.H) + 2 * y + (bz >> 1)) * (2 * p.Wd) + 2 * x + (bz & 1); }
            uint32_t ooff[TMW], roff[TMW];
            int pb = 0, py = 0, px = 0;
#pragma unroll
            for (int i = 0; i < TMW; ++i) {
                const int dm = (wm * TMW + i) * 16 + r16, m = m0 + dm;
                int64_t dmo = dm;
                if (phase) {
                    if (i == 0 || p.Wd < 16) { pb = m / hw; const int rem = m - pb * hw; py = rem / p.Wd; px = rem - py * p.Wd; }
                    else { px += 16; if (px >= p.Wd) { px -= p.Wd; if (++py == p.H) { py = 0; ++pb; } } }
                    dmo = ((int64_t)pb * (2 * p.H) + 2 * py + (bz >> 1)) * (2 * p.Wd) + 2 * px + (bz & 1) - mo0;
                }
                ooff[i] = ((uint32_t)dmo * (uint32_t)p.ldo + (uint32_t)col) * 4u;
                roff[i] = res ? ((uint32_t)dm * (uint32_t)p.ldr + (uint32_t)col) * 4u : 0u;
            }
            char* ob = (char*)(Ob + mo0 * p.ldo);
            const char* rb = res ? (const char*)(p.resid + (int64_t)m0 * p.ldr) : nullptr;
#pragma unroll
            for (int j = 0; j < TNW; ++j) {
                const f32x4 b4 = *(const f32x4*)(p.bias + col + j * 16);
                double gs[4] = {0.0, 0.0, 0.0, 0.0}, gq[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int i = 0; i < TMW; ++i) {
                    f32x4 v = acc[i][j] + b4;
                    if (res) v = *(const f32x4*)(rb + (uint64_t)roff[i] + j * 64) + v;
                    *(f32x4*)(ob + (uint64_t)ooff[i] + j * 64) = v;
                    if (p.gn_part) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) { const double d = (double)v[e]; gs[e] += d; gq[e] += d * d; }
                    }
                }
                if (p.gn_part) {                                        // as in the general loop: butterfly over the 16 pixel lanes, park in LDS
#pragma unroll
                    for (int off = 8; off >= 1; off >>= 1)
#pragma unroll
                        for (int e = 0; e < 4; ++e) { gs[e] += __shfl_xor(gs[e], off, 64); gq[e] += __shfl_xor(gq[e], off, 64); }
                    if (r16 == 0) {
                        double* red = reinterpret_cast<double*>(smem);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int nl = (wn * TNW + j) * 16 + kq * 4 + e;
                            red[(wm * BN + nl) * 2] = gs[e]; red[(wm * BN + nl) * 2 + 1] = gq[e];
                        }
                    }
                }
            }
            lean_done = true;
        }
    }
    if (!lean_done) {
#pragma unroll
    for (int j = 0; j < TNW; ++j) {
        const int n = nw0 + j * 16 + kq * 4;
        if (n >= p.N) continue;
        const bool full = p.evec && (n + 3 < p.N);
        f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
        if (p.bias && !p.bias_per_row) {
            if (full) b4 = *(const f32x4*)(p.bias + n);
            else { for (int e = 0; e < 4; ++e) if (n + e < p.N) b4[e] = p.bias[n + e]; }
        }
        double gs[4] = {0.0, 0.0, 0.0, 0.0}, gq[4] = {0.0, 0.0, 0.0, 0.0};     // GroupNorm partials of this lane's 4 channels (CONV, gn_part)
#pragma unroll
        for (int i = 0; i < TMW; ++i) {
            const int m = m0 + (wm * TMW + i) * 16 + r16;
            if (m >= p.M) continue;
            f32x4 v = acc[i][j];
            int64_t mo = m;                                                 // output row; phase mode scatters to the 2x grid
            if (CONV) {
                const int hw = p.H * p.Wd, b = m / hw, rem2 = m - b * hw;
                if (p.out_mode != 0) {                                      // last conv: NCHW store of <= 3 channels, clamp (+ de-normalise)
                    for (int e = 0; e < 4; ++e) {
                        if (n + e >= p.N) break;
                        const float x = vm_min(vm_max(v[e] + b4[e], -1.0f), 1.0f);
                        Ob[((int64_t)b * p.N + n + e) * hw + rem2] = p.out_mode == 1 ? (x + 1.0f) * 0.5f : x;
                    }
                    continue;
                }
                if (p.up2 == 2) {
                    const int y = rem2 / p.Wd, x = rem2 - y * p.Wd;
                    mo = ((int64_t)b * (2 * p.H) + 2 * y + (bz >> 1)) * (2 * p.Wd) + 2 * x + (bz & 1);
                }
            }
            if (p.bias) { if (p.bias_per_row) { const float bm = p.bias[m]; v[0] = v[0] + bm; v[1] = v[1] + bm; v[2] = v[2] + bm; v[3] = v[3] + bm; }
                          else v = v + b4; }
            if (p.epi == VARHIP_EPI_GELU) { v[0] = vm_gelu_tanh(v[0]); v[1] = vm_gelu_tanh(v[1]); v[2] = vm_gelu_tanh(v[2]); v[3] = vm_gelu_tanh(v[3]); }
            if (full) {
                if (p.epi == VARHIP_EPI_RESID) {
                    if (p.gamma) v = v * *(const f32x4*)(p.gamma + (int64_t)(m / p.rows_per_group) * p.ldg + n);
                    v = *(const f32x4*)(p.resid + (int64_t)m * p.ldr + n) + v;
                }
                *(f32x4*)(Ob + mo * p.ldo + n) = v;
                if (CONV && p.gn_part) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const double d = (double)v[e]; gs[e] += d; gq[e] += d * d; }
                }
            } else {
                for (int e = 0; e < 4; ++e) {
                    if (n + e >= p.N) break;
                    float x = v[e];
                    if (p.epi == VARHIP_EPI_RESID) {
                        if (p.gamma) x = x * p.gamma[(int64_t)(m / p.rows_per_group) * p.ldg + n + e];
                        x = p.resid[(int64_t)m * p.ldr + n + e] + x;
                    }
                    Ob[mo * p.ldo + n + e] = x;
                }
            }
        }
        if constexpr (CONV && NST == 2) {
            if (p.gn_part) {
                // sum over the wave's 16 pixel lanes (fixed butterfly), then park the 4 channel sums of this (wm, j, kq) in LDS: the
                // K-loop stages are free, the loop ended on a barrier.  The host guarantees full tiles (M % BM == 0, N % BN == 0).
#pragma unroll
                for (int off = 8; off >= 1; off >>= 1)
#pragma unroll
                    for (int e = 0; e < 4; ++e) { gs[e] += __shfl_xor(gs[e], off, 64); gq[e] += __shfl_xor(gq[e], off, 64); }
                if (r16 == 0) {
                    double* red = reinterpret_cast<double*>(smem);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int nl = (wn * TNW + j) * 16 + kq * 4 + e;
                        red[(wm * BN + nl) * 2] = gs[e]; red[(wm * BN + nl) * 2 + 1] = gq[e];
                    }
                }
            }
        }
    }
    }
    if constexpr (CONV && NST == 2) {
        if (p.gn_part) {
            __syncthreads();
            if (tid < BN) {
                const double* red = reinterpret_cast<const double*>(smem);
                const double s = red[tid * 2] + red[(BN + tid) * 2], q = red[tid * 2 + 1] + red[(BN + tid) * 2 + 1];
                const int hw = p.H * p.Wd, b = m0 / hw, per = hw / BM, blk = bz * per + (m0 - b * hw) / BM, nblk = (int)gridDim.z * per;
                double* o = p.gn_part + (((int64_t)b * nblk + blk) * p.N + n0 + tid) * 2;
                o[0] = s; o[1] = q;
            }
        }
    }
}

template <int TMW, int TNW, bool CONV = false, int NST = 2, bool GATHER = false>
static int launch_dma(GemmP& p, int batch, hipStream_t stream) {
    constexpr int BM = TMW * 32, BN = TNW * 32;
    constexpr size_t lds = NST * (size_t)(BM + BN) * 32 * sizeof(float);
    p.tilesM = (p.M + BM - 1) / BM; p.tilesN = (p.N + BN - 1) / BN;
    auto kfn = k_dma_gemm<TMW, TNW, CONV, NST, GATHER>;
    static bool attr_done = false;
    if (!attr_done) {
        if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    dim3 grid(p.tilesM * p.tilesN, 1, batch);
    hipLaunchKernelGGL(kfn, grid, dim3(256), lds, stream, p);
    return vh_launch_status();
}

static int launch_any(GemmP& p, int batch, hipStream_t stream) {
    p.tilesM = (p.M + 63) / 64; p.tilesN = (p.N + 63) / 64;
    hipLaunchKernelGGL(k_gemm_any, dim3(p.tilesM * p.tilesN, 1, batch), dim3(256), 0, stream, p);
    return vh_launch_status();
}

extern "C" int varhip_gemm_nt_f32(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias,
                                  float* out, int64_t ldo, int M, int N, int K, int epi,
                                  const float* resid, int64_t ldr, const float* gamma, int64_t ldg, int rows_per_group,
                                  int bias_per_row, int batch, int64_t sA, int64_t sW, int64_t sO, varhip_stream_t stream) {
    if (M < 0 || N <= 0 || K <= 0 || batch < 1) return VARHIP_EINVAL;
    // the DMA kernel addresses a lane's row as a 32-bit byte offset from the (per-batch) operand base
    const bool fits32 = ((int64_t)(M > 0 ? M - 1 : 0) * lda + K) * 4 < (1ll << 32) && ((int64_t)(N - 1) * ldw + K) * 4 < (1ll << 32);
    const bool vec = fits32 && !((K & 31) || (lda & 3) || (ldw & 3) || (sA & 3) || (sW & 3) || (((uintptr_t)A | (uintptr_t)W) & 15));
    if (batch > 1 && (resid || gamma)) return VARHIP_EINVAL;
    if (epi < 0 || epi > 2 || (epi == VARHIP_EPI_RESID && !resid)) return VARHIP_EINVAL;
    if (M == 0) return 0;
    GemmP p{};
    p.A = A; p.W = W; p.bias = bias; p.out = out; p.resid = resid; p.gamma = gamma;
    p.lda = lda; p.ldw = ldw; p.ldo = ldo; p.ldr = ldr; p.ldg = ldg; p.sA = sA; p.sW = sW; p.sO = sO;
    p.M = M; p.N = N; p.K = K; p.epi = epi; p.rows_per_group = rows_per_group > 0 ? rows_per_group : 1; p.bias_per_row = bias_per_row;
    p.evec = !((N & 3) || (ldo & 3) || (sO & 3) || ((uintptr_t)out & 15) || (bias && !bias_per_row && ((uintptr_t)bias & 15)) ||
               (resid && ((ldr & 3) || ((uintptr_t)resid & 15))) || (gamma && ((ldg & 3) || ((uintptr_t)gamma & 15))));
    // Tile choice.  K cannot be split (arithmetic contract), so launches that do not fill 256 CUs several times over pay for
    // block-count quantisation.  cost ~ quantised block count * tile area / relative tile efficiency; the block count is rounded
    // up to whole rounds of 256 CUs and (weight 1/4) to whole rounds of the co-resident workgroups of the tile (2 / 3 / 4 per CU
    // by LDS and registers) — fitted to the forced-tile timings of every d16 shape (tools/bench_kernels.py gemm with
    // VARHIP_GEMM_TILE=0/1/2): the picks land within 0.1 % of the per-shape best in total.
    auto cost = [&](int bm, int bn, int occ, double eff) {
        const int64_t nb = (int64_t)((M + bm - 1) / bm) * ((N + bn - 1) / bn) * batch, slots = 256 * occ;
        const double q = 0.75 * (double)((nb + 255) / 256 * 256) + 0.25 * (double)((nb + slots - 1) / slots * slots);
        return q * bm * bn / eff;
    };
    // Relative tile efficiencies with the unrolled main loop: the smaller tiles run the K loop as fast as the 128x128 one
    // (1.0 / 1.0 / 0.985); what separates them is the epilogue — a GELU or gamma*x+residual epilogue after a short K loop
    // costs the big tile ~5 % because fewer workgroups share a CU to overlap it (proj at l=256: 593 us vs 542 us).
    const double e128 = (epi != VARHIP_EPI_NONE && K <= 2048) ? 0.95 : 1.0;
    const double c128 = cost(128, 128, 2, e128), c12864 = cost(128, 64, 3, 1.0), c64 = cost(64, 64, 4, 0.985);
    int pick = !vec ? 3 : (c128 <= c12864 && c128 <= c64) ? 0 : (c12864 <= c64 ? 1 : 2);
    static const int forced = [] { const char* e = getenv("VARHIP_GEMM_TILE"); return e ? atoi(e) : -1; }();   // experiments only
    // fewer 64x64 tiles than half the CUs (the l = 1 and l = 4 scales): the lone workgroup of a CU is bound by one LDS round trip
    // per k-step and one memory latency per K tile, so 32x32 tiles (4x the workgroups) with 7 tiles of DMA in flight win
    if (pick == 2 && (int64_t)((M + 63) / 64) * ((N + 63) / 64) * batch <= 128) pick = 4;
    if (vec && forced >= 0 && forced <= 2) pick = forced;
    if (vec && forced == 3) pick = 4;
    VhScope scope(pick == 0 ? VH_FAM_GEMM : VH_FAM_GEMM_SMALL, (hipStream_t)stream, 2.0 * M * N * (double)K * batch,
                  4.0 * batch * ((double)M * K + (double)N * K + (double)M * N));
    switch (pick) {
        case 0: return launch_dma<4, 4>(p, batch, (hipStream_t)stream);
        case 1: return launch_dma<4, 2>(p, batch, (hipStream_t)stream);
        case 2: return launch_dma<2, 2>(p, batch, (hipStream_t)stream);
        case 4: return launch_dma<1, 1, false, 8>(p, batch, (hipStream_t)stream);
        default: return launch_any(p, batch, (hipStream_t)stream);                 // any K / alignment: scalar guarded loads
    }
}

// tile choice of the implicit-GEMM convolutions: the N tile divides Cout (160/320/640 -> 160 wide); Cin % 32 == 0 so that a
// K tile of 32 lies inside one tap
static int conv_family(int up2, int Cout) { return (up2 != 1 && Cout % 160 == 0) ? VH_FAM_CONV : VH_FAM_CONV_SMALL; }   // as launch_conv picks
static int launch_conv(GemmP& p, int batch, hipStream_t s) {
    if ((int64_t)p.N * p.ldw * 4 >= (1ll << 32)) return VARHIP_EINVAL;      // weight rows are 32-bit DMA offsets from the (per-phase) base
    {   // the input window of one workgroup's buffer descriptor (the samples a 128-pixel tile can touch + one row) must stay < 2 GB
        const int64_t hw = (int64_t)p.H * p.Wd, sample = (int64_t)p.Hi * p.Wi * p.Cin;
        if (p.up2 != 1 && ((127 / hw + 2) * sample + (int64_t)(p.Wi + 1) * p.Cin) * 4 >= (1ll << 31)) return VARHIP_EINVAL;
    }
    if (p.up2 == 1) {                                                 // nearest-2x gather (not on the hot path)
        if (p.N % 160 == 0) return launch_dma<4, 5, true, 2, true>(p, batch, s);
        if (p.N % 64 == 0) return launch_dma<4, 2, true, 2, true>(p, batch, s);
        return launch_dma<4, 1, true, 2, true>(p, batch, s);
    }
    if (p.N % 160 == 0) return launch_dma<4, 5, true>(p, batch, s);     // (64x160 tiles were measured: 6 % slower)
    if (p.N % 128 == 0) return launch_dma<4, 4, true>(p, batch, s);
    if (p.N % 64 == 0) return launch_dma<4, 2, true>(p, batch, s);
    return launch_dma<4, 1, true>(p, batch, s);         // (a 4-stage pipeline was measured here: slower, occupancy matters more)
}

// ---- mat_qkv with the q/k/v post-processing in the epilogue (basic_var.py:93-109): the [M][3C] intermediate never reaches HBM.
// Arithmetic identical to varhip_gemm_nt_f32 followed by varhip_qkv_prep_f32 (same fma chains, same W64 butterfly).
extern "C" int varhip_gemm_qkv_f32(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias, int M, int C, int K,
                                   const float* scale_mul, float plain_scale, int l2norm,
                                   float* q_out, float* kcache, float* vcache, int B2, int l, int H, int pos0, int Lmax,
                                   varhip_stream_t stream) {
    if (B2 <= 0 || l <= 0 || H <= 0 || pos0 < 0 || pos0 + l > Lmax || (l2norm && !scale_mul)) return VARHIP_EINVAL;
    if (C != H * 64 || M != B2 * l || K <= 0 || (K & 31) || (lda & 3) || (ldw & 3)) return VARHIP_EINVAL;
    if (((int64_t)(M - 1) * lda + K) * 4 >= (1ll << 32) || ((int64_t)(3 * C - 1) * ldw + K) * 4 >= (1ll << 32)) return VARHIP_EINVAL;   // 32-bit DMA offsets
    if ((((uintptr_t)A | (uintptr_t)W | (uintptr_t)bias | (uintptr_t)q_out | (uintptr_t)kcache | (uintptr_t)vcache) & 15)) return VARHIP_EINVAL;
    GemmP p{};
    p.A = A; p.W = W; p.bias = bias; p.lda = lda; p.ldw = ldw;
    p.M = M; p.N = 3 * C; p.K = K; p.epi = 3; p.rows_per_group = 1;
    p.q_smul = scale_mul; p.q_out = q_out; p.q_kc = kcache; p.q_vc = vcache; p.q_plain = plain_scale;
    p.q_l2 = l2norm; p.q_l = l; p.q_pos0 = pos0; p.q_Lmax = Lmax;
    auto cost = [&](int bm, double eff) {
        const int64_t nb = (int64_t)((M + bm - 1) / bm) * ((3 * C + 127) / 128);
        return (double)((nb + 255) / 256) * bm / eff;
    };
    static const int forced = [] { const char* e = getenv("VARHIP_QKV_TILE"); return e ? atoi(e) : -1; }();   // experiments only: 0 = 128 rows, 1 = 64
    const bool big = forced >= 0 ? forced == 0 : cost(128, 0.97) <= cost(64, 1.0);      // measured with the fused epilogue: the 64-row tile (3 workgroups per CU) is level or ahead at every d16 scale
    VhScope scope(big ? VH_FAM_GEMM : VH_FAM_GEMM_SMALL, (hipStream_t)stream, 2.0 * M * 3.0 * C * (double)K,
                  4.0 * ((double)M * K + 3.0 * C * K + 3.0 * M * C));
    return big ? launch_dma<4, 4>(p, 1, (hipStream_t)stream) : launch_dma<2, 4>(p, 1, (hipStream_t)stream);
}

// GroupNorm partials from the conv epilogue: blocks of 128 consecutive pixels never straddle samples and tiles are full
extern "C" int varhip_conv_gn_blocks(int H, int W, int Cout, int phase) {
    const int hw = phase ? (H / 2) * (W / 2) : H * W;
    if (H <= 0 || W <= 0 || Cout <= 0 || (Cout & 31) || (hw & 127) || (phase && ((H & 1) || (W & 1)))) return 0;
    return (phase ? 4 : 1) * (hw / 128);
}

static int conv3x3_impl(const float* in, const float* w, const float* bias, const float* resid, float* out, double* gn_part,
                        int B, int H, int W, int Cin, int Cout, int up2, int out_mode, varhip_stream_t stream) {
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || (Cin & 31) || !bias) return VARHIP_EINVAL;
    if (gn_part && (out_mode != 0 || !varhip_conv_gn_blocks(H, W, Cout, 0))) return VARHIP_EINVAL;
    if (up2 && ((H & 1) || (W & 1))) return VARHIP_EINVAL;
    if (out_mode < 0 || out_mode > 2 || (out_mode != 0 && resid)) return VARHIP_EINVAL;
    if ((int64_t)B * H * W >= (1ll << 31)) return VARHIP_EINVAL;
    GemmP p{};
    p.A = in; p.W = w; p.bias = bias; p.out = out; p.resid = resid; p.gamma = nullptr;
    p.ldw = 9ll * Cin; p.ldo = Cout; p.ldr = Cout;
    p.M = B * H * W; p.N = Cout; p.K = 9 * Cin; p.epi = resid ? VARHIP_EPI_RESID : VARHIP_EPI_NONE; p.rows_per_group = 1;
    p.H = H; p.Wd = W; p.Cin = Cin; p.up2 = up2; p.out_mode = out_mode; p.Hi = up2 ? H / 2 : H; p.Wi = up2 ? W / 2 : W;
    p.evec = !((Cout & 3) || ((uintptr_t)out & 15) || ((uintptr_t)bias & 15) || (resid && ((uintptr_t)resid & 15)));
    if (gn_part && !p.evec) return VARHIP_EINVAL;
    p.gn_part = gn_part;
    const double npix = (double)B * H * W;
    VhScope scope(conv_family(up2, Cout), (hipStream_t)stream, 2.0 * npix * Cout * 9.0 * Cin,
                  4.0 * (npix * Cin / (up2 ? 4.0 : 1.0) + npix * Cout * (resid ? 2.0 : 1.0) + 9.0 * Cin * Cout));
    hipStream_t s = (hipStream_t)stream;
    return launch_conv(p, 1, s);
}
extern "C" int varhip_conv3x3_nhwc_f32(const float* in, const float* w, const float* bias, const float* resid, float* out,
                                       int B, int H, int W, int Cin, int Cout, int up2, int out_mode, varhip_stream_t stream) {
    return conv3x3_impl(in, w, bias, resid, out, nullptr, B, H, W, Cin, Cout, up2, out_mode, stream);
}
extern "C" int varhip_conv3x3_gn_nhwc_f32(const float* in, const float* w, const float* bias, const float* resid, float* out, double* gn_part,
                                          int B, int H, int W, int Cin, int Cout, int up2, varhip_stream_t stream) {
    if (!gn_part) return VARHIP_EINVAL;
    return conv3x3_impl(in, w, bias, resid, out, gn_part, B, H, W, Cin, Cout, up2, 0, stream);
}

// ---- Downsample2x of the encoder (basic_vae.py:31-37): F.pad(x, (0,1,0,1)) then Conv2d(k=3, stride=2, padding=0) -------------
extern "C" int varhip_conv3x3_s2_nhwc_f32(const float* in, const float* w, const float* bias, float* out,
                                          int B, int H, int W, int Cin, int Cout, varhip_stream_t stream) {
    // in: [B][2H][2W][Cin], out: [B][H][W][Cout]
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || (Cin & 31) || !bias) return VARHIP_EINVAL;
    if ((int64_t)B * H * W >= (1ll << 31)) return VARHIP_EINVAL;
    GemmP p{};
    p.A = in; p.W = w; p.bias = bias; p.out = out;
    p.ldw = 9ll * Cin; p.ldo = Cout; p.ldr = Cout;
    p.M = B * H * W; p.N = Cout; p.K = 9 * Cin; p.epi = VARHIP_EPI_NONE; p.rows_per_group = 1;
    p.H = H; p.Wd = W; p.Cin = Cin; p.up2 = 3; p.out_mode = 0; p.Hi = 2 * H; p.Wi = 2 * W;
    p.evec = !((Cout & 3) || ((uintptr_t)out & 15) || ((uintptr_t)bias & 15));
    const double npix = (double)B * H * W;
    VhScope scope(conv_family(3, Cout), (hipStream_t)stream, 2.0 * npix * Cout * 9.0 * Cin, 4.0 * (npix * 4 * Cin + npix * Cout + 9.0 * Cin * Cout));
    hipStream_t s = (hipStream_t)stream;
    return launch_conv(p, 1, s);
}

// ---- nearest-2x upsample + 3x3 conv as four 2x2 convs on the low-resolution map ("phase decomposition") ---------------------
// For output pixel (2y+py, 2x+px) the three kernel rows hit only two low-res rows (py=0: y-1 | y,y ; py=1: y,y | y+1), same for
// columns, so the taps that share a source pixel are pre-summed: 4 taps instead of 9 -> 2.25x fewer MACs than the reference's
// F.interpolate(nearest) + conv (basic_vae.py:27-28).  Pre-summing weights changes rounding (~1e-7 relative): allowed, the decoder
// is off the token path (pixels within 1e-3); the oracle keeps the plain 9-tap definition.
__global__ void k_upconv_pack(const float* __restrict__ w, float* __restrict__ wp, int Cout, int Cin) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;          // wp: [4 phases][Cout][2][2][Cin]
    const int64_t tot = (int64_t)16 * Cout * Cin;
    if (i >= tot) return;
    const int ci = (int)(i % Cin); int64_t t = i / Cin; const int b = (int)(t & 1), a = (int)((t >> 1) & 1); t >>= 2;
    const int co = (int)(t % Cout), ph = (int)(t / Cout), py = ph >> 1, px = ph & 1;
    const int ky0 = (py == 0) ? (a == 0 ? 0 : 1) : (a == 0 ? 0 : 2), ky1 = (py == 0) ? (a == 0 ? 0 : 2) : (a == 0 ? 1 : 2);
    const int kx0 = (px == 0) ? (b == 0 ? 0 : 1) : (b == 0 ? 0 : 2), kx1 = (px == 0) ? (b == 0 ? 0 : 2) : (b == 0 ? 1 : 2);
    float s = 0.f;
    for (int ky = ky0; ky <= ky1; ++ky)
        for (int kx = kx0; kx <= kx1; ++kx) s = s + w[(((int64_t)co * 3 + ky) * 3 + kx) * Cin + ci];
    wp[i] = s;
}
extern "C" int varhip_upconv_pack_f32(const float* w, float* w_phase, int Cin, int Cout, varhip_stream_t stream) {
    if (Cin <= 0 || Cout <= 0) return VARHIP_EINVAL;
    const int64_t tot = (int64_t)16 * Cout * Cin;
    VhScope scope(VH_FAM_OTHER, (hipStream_t)stream, 0, 4.0 * tot);
    hipLaunchKernelGGL(k_upconv_pack, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, w_phase, Cout, Cin);
    return vh_launch_status();
}

static int upconv_phase_impl(const float* in, const float* w_phase, const float* bias, float* out, double* gn_part,
                             int B, int H, int W, int Cin, int Cout, varhip_stream_t stream) {
    // in: [B][H/2][W/2][Cin]; out: [B][H][W][Cout]; w_phase from varhip_upconv_pack_f32
    if (B <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1) || Cin <= 0 || Cout <= 0 || (Cin & 31) || !bias) return VARHIP_EINVAL;
    if (gn_part && !varhip_conv_gn_blocks(H, W, Cout, 1)) return VARHIP_EINVAL;
    if ((int64_t)B * H * W >= (1ll << 31)) return VARHIP_EINVAL;
    GemmP p{};
    p.A = in; p.W = w_phase; p.bias = bias; p.out = out; p.resid = nullptr; p.gamma = nullptr;
    p.ldw = 4ll * Cin; p.ldo = Cout; p.ldr = Cout; p.sW = (int64_t)Cout * 4 * Cin;
    p.M = B * (H / 2) * (W / 2); p.N = Cout; p.K = 4 * Cin; p.epi = VARHIP_EPI_NONE; p.rows_per_group = 1;
    p.H = H / 2; p.Wd = W / 2; p.Cin = Cin; p.up2 = 2; p.out_mode = 0; p.Hi = H / 2; p.Wi = W / 2;
    p.evec = !((Cout & 3) || ((uintptr_t)out & 15) || ((uintptr_t)bias & 15));
    if (gn_part && !p.evec) return VARHIP_EINVAL;
    p.gn_part = gn_part;
    const double npix = (double)B * H * W;
    VhScope scope(conv_family(2, Cout), (hipStream_t)stream, 2.0 * npix * Cout * 4.0 * Cin, 4.0 * (npix * Cin / 4.0 + npix * Cout + 16.0 * Cin * Cout));
    hipStream_t s = (hipStream_t)stream;
    return launch_conv(p, 4, s);
}
extern "C" int varhip_upconv_phase_f32(const float* in, const float* w_phase, const float* bias, float* out,
                                       int B, int H, int W, int Cin, int Cout, varhip_stream_t stream) {
    return upconv_phase_impl(in, w_phase, bias, out, nullptr, B, H, W, Cin, Cout, stream);
}
extern "C" int varhip_upconv_phase_gn_f32(const float* in, const float* w_phase, const float* bias, float* out, double* gn_part,
                                          int B, int H, int W, int Cin, int Cout, varhip_stream_t stream) {
    if (!gn_part) return VARHIP_EINVAL;
    return upconv_phase_impl(in, w_phase, bias, out, gn_part, B, H, W, Cin, Cout, stream);
}
