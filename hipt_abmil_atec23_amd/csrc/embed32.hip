// PATCH EMBEDDING of ViT-256 straight from the fp32 image:   x[s, 1 + t, :] = W . pixels(s, t) + b + pos[1 + t]
//   (PatchEmbed.forward = Conv2d(3, 384, k = 16, s = 16) + flatten, HIPT_4K/vision_transformer.py:155-170, fused with the
//   unfold / rearrange patchify of hipt_4k.py:64-65 through hipt_image_layout and with `x + pos` of prepare_tokens :235-246.)
//
// Round 1 made a bf16 copy of the image (f32_to_bf16: 2.4 GB read + written per 8 regions) and ran the generic LDS-tiled GEMM
// over an im2col view of it, whose 16-pixel runs are 32-byte pieces: 6.2 ms per 24-region step, 7 % of it.  Here a workgroup owns
// 128 tokens -- 8 token rows of one 256 x 256 patch, i.e. 128 image rows x 256 contiguous pixels per channel -- and works like the
// fc2 half of mlp32.hip:
//   * a wave's 32 tokens (2 token rows x 16) are ONE MFMA B operand: lane l = 32 h + 16 m + li is token (row m, column li) and holds,
//     for k-step (channel c, pixel row r), the 8 pixels 8 h .. 8 h + 7 of that row -- two 16-byte fp32 loads, rounded to bf16 in
//     registers.  A load instruction covers whole 64-byte lines; the image is read once, as fp32.
//   * the weight [384, 768] streams through the 3 x 48 KiB LDS-DMA ring as A operand fragments (32 outputs x 16 k = 1 KiB,
//     conflict-free by construction), 12 units per tile pass = 3 channels x 4; the [32 tokens x 384] accumulator (192 registers)
//     takes all of them;
//   * the pixels of a channel are fetched two channels ahead, a quarter (4 pixel rows) per phase, into one of three operand
//     buffers (3 x 64 registers), across tile boundaries.  vmcnt is ONE in-order counter for these loads and the DMA pieces:
//     a phase requests its quarter AFTER its last DMA piece (group 5) and waits with vmcnt(8) at its end -- the pieces have
//     landed, the eight pixel loads may still be in flight; they are rounded to bf16 in the middle of the NEXT phase.
//   * epilogue: + bias + pos, fp32 rows 1 + t of the sequence (row 0, the [CLS] slot, is written by cls_init).
//   * KIND 1 / 2: the image is uint8 RGB, planar or interleaved (what a decoded slide tile is: a quarter of the bytes over PCIe
//     and HBM).  ToTensor + Normalize(0.5, 0.5) happen in registers: a lane's 8 pixels of a row are 8 consecutive bytes (planar) or,
//     with all three channels, 24 (interleaved: every channel's phase re-reads the run and keeps its own bytes);
//     bf16((b / 255 - 0.5) / 0.5) -- the reference's arithmetic, hipt_model_utils.py:113-118 -- equals bf16(fma(b, 2/255, -1)) for every one of
//     the 256 byte values with 2/255 rounded to 0x3c008081 (tests/test_host_and_abi.py checks all of them), so the two divisions
//     become one v_cvt_f32_ubyte + one v_fma_f32 per pixel and the tokens are the bits of the float path.
#include <stdio.h>
#include <stdlib.h>

#include "common.h"
#include "kernels.h"
#include "pipe_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int D = 384, NOT = 12, NCHN = 3, KPC = 256;  // KPC: k per channel = 16 pixel rows x 16 pixels
constexpr int UNIT = 48 * 1024, UPT = 12;                          // ring unit = one phase = 48 fragments; units per tile pass

__device__ __forceinline__ void mma32(f32x16& acc, const u32x4& a, const u32x4& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
}

// The packed image: unit pos = 4 c + u (channel c, u = 0..3), fragment f = 4 gg + t of group gg (0..11): output tile O = 3 u + gg / 4,
// k-step r = 4 (gg % 4) + t (pixel row of the channel); lane (j = l & 31, h = l >> 5): W[32 O + j][256 c + 16 r + 8 h + (0..7)].
__global__ void embed32_pack_kernel(const bf16_t* __restrict__ w, u32x4* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;  // one 16-byte lane chunk
    if (i >= UPT * (UNIT / 16)) return;
    const int pos = i / (UNIT / 16), o = i % (UNIT / 16), frag = o >> 6, lane = o & 63, j = lane & 31, h = lane >> 5;
    const int c = pos >> 2, u = pos & 3, gg = frag >> 2, t = frag & 3, O = 3 * u + (gg >> 2), r = 4 * (gg & 3) + t;
    out[i] = *(const u32x4*)(w + (int64_t)(32 * O + j) * (NCHN * KPC) + KPC * c + 16 * r + 8 * h);
}

// LNOUT (round 5): the token rows leave as ACTIVATION IMAGES (kernels.h) -- x as the fp32 image and LayerNorm-1 of the first block applied to it
// as the bf16 image p.xn_out -- so that the first block runs the kernels of all the others (fused QKV + attention, image-in fused MLP) instead
// of LayerNorm-in-GEMM + a q | k | v tensor + the two-kernel attention.  A token's 384 values are the 192 of this lane and the 192 of lane ^ 32.
template <int KIND, bool LNOUT>
__global__ __launch_bounds__(256, 1) void embed32_kernel(const EmbedParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* bs = (float*)(smem + 3 * UNIT);  // bias [D]
    float* gs = bs + D;                     // LNOUT: gamma [D] | beta [D] of the first block's LayerNorm-1
    int* tile_s = (int*)(bs + (LNOUT ? 3 : 1) * D);  // [2] tile handed to this workgroup, double-buffered by parity

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, h = lane >> 5, m = (lane >> 4) & 1;  // lane = 32 h + 16 m + li: token (row m, column li), k half h
    const int tps = p.nty / 8;                                     // tiles per sequence

    // ---- weight DMA: unit pos of the image = 48 pieces of 1 KiB; wave w issues pieces 12 w .. 12 w + 11, four per M0 value ----
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.wpk, 0, UPT * UNIT, 0x00020000);
    const uint32_t ilane = (uint32_t)(12 * wave * 1024 + lane * 16);
    int ioff = 0, islot = 0, ipos = 0;
    auto set_issue = [&](int pos, int slot) {
        ioff = pos * UNIT;
        islot = slot;
    };
    auto dma_piece = [&](auto T_) __attribute__((always_inline)) {
        constexpr int t = decltype(T_)::value;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (LDS_AS void*)(smem + islot * UNIT + (12 * wave + (t & ~3)) * 1024), 16, ilane, ioff + (t & ~3) * 1024, (t & 3) * 1024, 0);
    };

    for (int i = tid; i < D; i += 256) bs[i] = p.bias[i];
    if constexpr (LNOUT)
        for (int i = tid; i < D; i += 256) {
            gs[i] = p.ln_w[i];
            gs[D + i] = p.ln_b[i];
        }
    if (tid == 0) tile_s[0] = atomicAdd(p.counter, 1);
    __syncthreads();
    int tile = __builtin_amdgcn_readfirstlane(tile_s[0]);

    const uint32_t lbase = (uint32_t)(uintptr_t)(LDS_AS char*)smem;
    const uint32_t fbase = lbase + lane * 16;                                       // + slot * UNIT + fragment * 1024
    const uint32_t bbase = (uint32_t)(uintptr_t)(LDS_AS char*)bs + 16 * h;          // bias[32 O + 8 q + 4 h ..]: + (32 O + 8 q) * 4
    const uint32_t tsbase = (uint32_t)(uintptr_t)(LDS_AS char*)tile_s;

    // this lane's pixel row 0, columns 8 h .. of channel 0 for tile t (hipt_image_layout: include/hipt_abmil.h), as an ELEMENT offset
    // (fp32 / planar uint8: elements of the tensor; interleaved: pixels -- 3 bytes each)
    typedef typename std::conditional<KIND == 0, float, uint8_t>::type pix_t;
    const pix_t* img = (const pix_t*)p.img;
    const int64_t plane = p.im.batch_stride / 3;  // (interleaved: pixels per image)
    auto pix_base = [&](int t) __attribute__((always_inline)) {
        const int b = p.seq0 + t / tps, ty = (t % tps) * 8 + 2 * wave + m;
        const int gsz = p.im.grid_w * p.im.grid_h, bi = b / gsz, s = b % gsz, p1 = s / p.im.grid_h, p2 = s % p.im.grid_h;
        const int64_t inimg = (int64_t)(p1 * p.im.patch_h + ty * 16) * p.im.row_stride + p2 * p.im.patch_w + li * 16 + 8 * h;
        if constexpr (KIND == 2) return img + ((int64_t)bi * plane + inimg) * 3;
        else return img + (int64_t)bi * p.im.batch_stride + inimg;
    };

    u32x4 X[NCHN][16];  // operand fragments: channel c, k-step r (pixel row): the lane's 8 pixels as bf16
    // a quarter of a channel in flight: pixel rows 4 j .. 4 j + 3.  fp32: two 16-byte pieces per row; uint8 planar: one 8-byte piece;
    // interleaved: the 24 bytes of the 8 pixels' three channels
    constexpr int RAWN = KIND == 0 ? 8 : (KIND == 1 ? 2 : 6);  // 16-byte registers
    // load INSTRUCTIONS per quarter = the counted wait of a phase (vmcnt(NLD): "my DMA pieces have landed, the NLD pixel loads behind them may still
    // fly").  The count must be the compiler's, not the source's: written as three 8-byte loads per row, the interleaved form was merged by hipcc
    // into one 16-byte + one 8-byte load -- 8 instructions where the wait said 12, so that the four youngest DMA pieces of the next ring unit could
    // still be in flight behind the barrier: a race on the weight ring that only showed as run-to-run differences when two streams embedded
    // uint8 regions at once (found in round 6 by the H2D loop's bit-equality test).  The loads are now written in the widths the hardware takes (16 + 8
    // bytes: nothing left to merge; `volatile` is no way out -- hipcc turns such loads into FLAT loads, which retire out of order), and
    // tools/audit_ring_waits.py re-counts them in every build's listing: a mismatch fails the build.
    constexpr int NLD = KIND == 0 ? 8 : (KIND == 1 ? 4 : 8);
    u32x4 raw[RAWN];
    auto load_quarter = [&](const pix_t* base, auto C_, auto J_) __attribute__((always_inline)) {
        constexpr int c = decltype(C_)::value, j = decltype(J_)::value;
        if constexpr (KIND == 0) {
            const float* q = (const float*)base + (int64_t)c * p.im.chan_stride + (int64_t)(4 * j) * p.im.row_stride;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                raw[2 * t] = *(const u32x4*)(q + (int64_t)t * p.im.row_stride);
                raw[2 * t + 1] = *(const u32x4*)(q + (int64_t)t * p.im.row_stride + 4);
            }
        } else if constexpr (KIND == 1) {
            const uint8_t* q = (const uint8_t*)base + (int64_t)c * p.im.chan_stride + (int64_t)(4 * j) * p.im.row_stride;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const u32x2 w = *(const u32x2*)(q + (int64_t)t * p.im.row_stride);
                raw[t >> 1][2 * (t & 1)] = w[0];
                raw[t >> 1][2 * (t & 1) + 1] = w[1];
            }
        } else {
            // the 24 bytes of a row's 8 pixels x 3 channels: one 16-byte + one 8-byte load (8-byte aligned: row_stride % 8 == 0)
            const uint8_t* q = (const uint8_t*)base + (int64_t)(4 * j) * p.im.row_stride * 3;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const u32x4 a = *(const u32x4*)(q + (int64_t)t * p.im.row_stride * 3);
                const u32x2 b = *(const u32x2*)(q + (int64_t)t * p.im.row_stride * 3 + 16);
                const uint32_t dw[6] = {a[0], a[1], a[2], a[3], b[0], b[1]};
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    const int d = 6 * t + k;  // dword index of the quarter's 24 dwords
                    raw[d >> 2][d & 3] = dw[k];
                }
            }
        }
    };
    auto cvt_quarter = [&](auto C_, auto J_) __attribute__((always_inline)) {
#pragma clang fp contract(off)
        constexpr int c = decltype(C_)::value, j = decltype(J_)::value;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            u32x4 o;
            if constexpr (KIND == 0) {
                const f32x4 r0 = __builtin_bit_cast(f32x4, raw[2 * t]), r1 = __builtin_bit_cast(f32x4, raw[2 * t + 1]);
                o[0] = pack_bf16x2(r0[0], r0[1]);
                o[1] = pack_bf16x2(r0[2], r0[3]);
                o[2] = pack_bf16x2(r1[0], r1[1]);
                o[3] = pack_bf16x2(r1[2], r1[3]);
            } else {
                // pixel i of row t: byte i of its 8 (planar) / byte 3 i + c of its 24 (interleaved)
                float f[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int byte = KIND == 1 ? 8 * t + i : 24 * t + 3 * i + c;  // within the quarter's bytes
                    const uint32_t w = raw[byte >> 4][(byte >> 2) & 3];
                    f[i] = __builtin_fmaf((float)((w >> (8 * (byte & 3))) & 0xffu), __builtin_bit_cast(float, 0x3c008081u), -1.0f);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = pack_bf16x2(f[2 * e], f[2 * e + 1]);
            }
            X[c][4 * j + t] = o;
        }
    };
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 3> I3;

    // ---- prime: ring units 0 and 1; channels 0 and 1 of the first tile (the only pixels this workgroup ever waits for) ----
    int cons = 0;  // units consumed since kernel start (slot = cons % 3)
    const pix_t* base_cur = img;
    if (tile < p.ntiles) {
        set_issue(0, 0);
        sfor<0, 12>(dma_piece);
        set_issue(1, 1);
        sfor<0, 2>(dma_piece);
        ipos = 2;
        base_cur = pix_base(tile);
        sfor<0, 2>([&](auto C_) __attribute__((always_inline)) {
            sfor<0, 4>([&](auto J_) __attribute__((always_inline)) {
                load_quarter(base_cur, C_, J_);
                cvt_quarter(C_, J_);
            });
        });
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    u32x4 wA[2][4];
    auto rd_frag = [&](auto SET_, auto G_, uint32_t sa) __attribute__((always_inline)) {
        constexpr int set = decltype(SET_)::value, gg = decltype(G_)::value;
        const uint32_t a = sa;
        u32x4 &d0 = wA[set][0], &d1 = wA[set][1], &d2 = wA[set][2], &d3 = wA[set][3];
        DSR128(d0, a, (4 * gg + 0) * 1024);
        DSR128(d1, a, (4 * gg + 1) * 1024);
        DSR128(d2, a, (4 * gg + 2) * 1024);
        DSR128(d3, a, (4 * gg + 3) * 1024);
    };

    for (int seq = 0; tile < p.ntiles; ++seq) {
        if (tid == 0) {  // next tile: fetched now, read after the first ring barrier
            const int nt = atomicAdd(p.counter, 1);
            asm volatile("ds_write_b32 %0, %1" ::"v"(tsbase + 4 * ((seq + 1) & 1)), "v"(nt) : "memory");
        }
        f32x16 acc[NOT];  // lane holds its token's output columns 32 O + 8 (reg >> 2) + 4 h + (reg & 3)
#pragma unroll
        for (int o = 0; o < NOT; ++o)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[o][e] = 0.f;
        const pix_t* base_next = base_cur;
        int next = p.ntiles;

        // ---- one phase: 12 groups of 4 MFMAs on the unit in slot cons % 3: channel C, unit U of it ----
        // LAST: last phase of the tile, nothing is prefetched across the row phase (the compiler moves registers there)
        auto phase = [&](auto C_, auto U_, auto LAST_, auto&& mid) __attribute__((always_inline)) {
            constexpr int c = decltype(C_)::value, u = decltype(U_)::value, last = decltype(LAST_)::value;
            const uint32_t sa = fbase + (cons % 3) * UNIT;
            const uint32_t sn = fbase + ((cons + 1) % 3) * UNIT;
            sfor<0, 12>([&](auto G_) __attribute__((always_inline)) {
                constexpr int gg = decltype(G_)::value, set = gg & 1;
                typedef std::integral_constant<int, set ^ 1> NS;
                if constexpr (gg == 11) {
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD) : "memory");  // my pieces of the next unit have landed (the pixel loads after them may not)
                    __builtin_amdgcn_s_barrier();                     // ... everyone's; unit cons-1 is no longer read
                    set_issue(ipos, (cons + 2) % 3);
                    ipos = ipos + 1 == UPT ? 0 : ipos + 1;
                    if constexpr (last) {
                        LGKM(0);
                    } else {
                        rd_frag(NS{}, I0{}, sn);
                        LGKM(4);
                    }
                } else {
                    rd_frag(NS{}, std::integral_constant<int, gg + 1>{}, sa);
                    LGKM(4);
                }
                constexpr int O = 3 * u + (gg >> 2), kq = gg & 3;
#pragma unroll
                for (int t = 0; t < 4; ++t) mma32(acc[O], wA[set][t], X[c][4 * kq + t]);
                if constexpr (gg == 11) {
                    dma_piece(std::integral_constant<int, 0>{});
                    dma_piece(std::integral_constant<int, 1>{});
                } else if constexpr (gg <= 4) {
                    dma_piece(std::integral_constant<int, 2 + 2 * gg>{});
                    dma_piece(std::integral_constant<int, 3 + 2 * gg>{});
                } else if constexpr (gg == 5) {
                    mid();
                }
                __builtin_amdgcn_sched_barrier(0);
            });
            cons += 1;
        };

        rd_frag(I0{}, I0{}, fbase + (cons % 3) * UNIT);
        // Phase (c, j), after its last DMA piece: round the quarter requested in the PREVIOUS phase, then request quarter j of the
        // channel two ahead (tc: channel 2 of this tile for c = 0, channel 0 / 1 of the next tile for c = 1 / 2).
        sfor<0, NCHN>([&](auto C_) __attribute__((always_inline)) {
            constexpr int c = decltype(C_)::value, tc = (c + 2) % NCHN;
            sfor<0, 4>([&](auto J_) __attribute__((always_inline)) {
                constexpr int j = decltype(J_)::value;
                if constexpr (c == 1 && j == 0) {
                    int nt;
                    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(nt) : "v"(tsbase + 4 * ((seq + 1) & 1)) : "memory");
                    next = __builtin_amdgcn_readfirstlane(nt);
                    if (next < p.ntiles) base_next = pix_base(next);  // (no next tile: the current tile's pixels again, never used)
                }
                phase(C_, J_, std::integral_constant<int, (c == NCHN - 1 && j == 3) ? 1 : 0>{}, [&]() __attribute__((always_inline)) {
                    if (seq > 0 || c > 0 || j > 0) {  // (the very first phase of a workgroup has no quarter in flight)
                        if constexpr (j > 0)
                            cvt_quarter(std::integral_constant<int, tc>{}, std::integral_constant<int, (j > 0 ? j - 1 : 0)>{});
                        else
                            cvt_quarter(std::integral_constant<int, (c + 1) % NCHN>{}, I3{});
                    }
                    load_quarter(c == 0 ? base_cur : base_next, std::integral_constant<int, tc>{}, J_);
                });
            });
        });

        // ---- epilogue: + bias + pos -> token rows of x.  acc[O][4 q + e] is output column 32 O + 8 q + 4 h + e of this lane's token ----
        if constexpr (!LNOUT) {
            const int b = tile / tps, t = ((tile % tps) * 8 + 2 * wave + m) * p.ntx + li;
            float* xr = p.x + ((int64_t)b * p.ntok + 1 + t) * D + 4 * h;
            const float* pr = p.pos + (int64_t)(1 + t) * D + 4 * h;
            sfor<0, NOT>([&](auto O_) __attribute__((always_inline)) {
                constexpr int O = decltype(O_)::value;
                f32x4 pv[4], bb[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) pv[q] = *(const f32x4*)(pr + 32 * O + 8 * q);
                const uint32_t ba = bbase;
                f32x4 &r0_ = bb[0], &r1_ = bb[1], &r2_ = bb[2], &r3_ = bb[3];
                DSR128X4_WAIT(r0_, r1_, r2_, r3_, ba, O * 128, O * 128 + 32, O * 128 + 64, O * 128 + 96);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (acc[O][4 * q + e] + bb[q][e]) + pv[q][e];
                    *(f32x4*)(xr + 32 * O + 8 * q) = v;
                }
            });
        } else {
#pragma clang fp contract(off)
            const int b = tile / tps, t = ((tile % tps) * 8 + 2 * wave + m) * p.ntx + li;
            const float* pr = p.pos + (int64_t)(1 + t) * D + 4 * h;
            // pass 1: the token values, in place; the row sum (this lane's 192 + the partner's)
            float sum = 0.f;
            sfor<0, NOT>([&](auto O_) __attribute__((always_inline)) {
                constexpr int O = decltype(O_)::value;
                f32x4 pv[4], bb[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) pv[q] = *(const f32x4*)(pr + 32 * O + 8 * q);
                const uint32_t ba = bbase;
                f32x4 &r0_ = bb[0], &r1_ = bb[1], &r2_ = bb[2], &r3_ = bb[3];
                DSR128X4_WAIT(r0_, r1_, r2_, r3_, ba, O * 128, O * 128 + 32, O * 128 + 64, O * 128 + 96);
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float v = (acc[O][4 * q + e] + bb[q][e]) + pv[q][e];
                        acc[O][4 * q + e] = v;
                        sum += v;
                    }
            });
            sum += __shfl_xor(sum, 32, 64);
            const float mean = sum / (float)D;
            float var = 0.f;
            sfor<0, NOT>([&](auto O_) __attribute__((always_inline)) {
                constexpr int O = decltype(O_)::value;
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const float d = acc[O][k] - mean;
                    var = __builtin_fmaf(d, d, var);
                }
            });
            var += __shfl_xor(var, 32, 64);
            const float rstd = 1.0f / sqrtf(var / (float)D + p.ln_eps);
            // pass 2: x as the fp32 image, LayerNorm-1(x) as the bf16 image: row R, 16-byte chunk 4 O + q (g = q, c = O of kernels.h), half h
            const int64_t R = (int64_t)b * p.ntok + 1 + t;
            float* xi = p.x + (R >> 4) * 6144 + h * 256 + (int)(R & 15) * 4;                       // + O * 512 + q * 64
            bf16_t* ni = (bf16_t*)p.xn_out + (R >> 4) * 6144 + (int)(R & 15) * 8 + 4 * h;           // + O * 512 + q * 128
            const uint32_t ga = bbase + D * 4, be = bbase + 2 * D * 4;
            sfor<0, NOT>([&](auto O_) __attribute__((always_inline)) {
                constexpr int O = decltype(O_)::value;
                f32x4 gg[4], bt[4];
                {
                    f32x4 &r0_ = gg[0], &r1_ = gg[1], &r2_ = gg[2], &r3_ = gg[3];
                    DSR128X4_WAIT(r0_, r1_, r2_, r3_, ga, O * 128, O * 128 + 32, O * 128 + 64, O * 128 + 96);
                }
                {
                    f32x4 &r0_ = bt[0], &r1_ = bt[1], &r2_ = bt[2], &r3_ = bt[3];
                    DSR128X4_WAIT(r0_, r1_, r2_, r3_, be, O * 128, O * 128 + 32, O * 128 + 64, O * 128 + 96);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 v;
                    float y[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = acc[O][4 * q + e];
                        y[e] = __builtin_fmaf((v[e] - mean) * rstd, gg[q][e], bt[q][e]);
                    }
                    *(f32x4*)(xi + O * 512 + q * 64) = v;
                    u32x2 o;
                    o[0] = pack_bf16x2(y[0], y[1]);
                    o[1] = pack_bf16x2(y[2], y[3]);
                    *(u32x2*)(ni + O * 512 + q * 128) = o;
                }
            });
        }
        tile = next;
        base_cur = base_next;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (continuous stream: pieces of a pass that never runs)
}

}  // namespace

bool hipt_embed32_supported(int dtype, int D_, int K, int nty, int ntx) {
    // (HIPT_GENERIC: the generic path = a bf16 copy of the image + the im2col GEMM)
    return !hipt_generic_only() && dtype == HIPT_BF16 && D_ == D && K == NCHN * KPC && ntx == 16 && nty > 0 && nty % 8 == 0;
}

size_t hipt_embed32_packed_bytes() { return (size_t)UPT * UNIT; }

int hipt_embed32_pack_launch(const void* w, void* packed, hipStream_t st) {
    const int chunks = UPT * (UNIT / 16);
    hipLaunchKernelGGL(embed32_pack_kernel, dim3((chunks + 255) / 256), dim3(256), 0, st, (const bf16_t*)w, (u32x4*)packed);
    HIPT_CHECK_LAUNCH();
    return HIPT_OK;
}

int hipt_embed32_launch(const EmbedParams& p_in, hipStream_t st) {
    EmbedParams p = p_in;
    HIPT_CHECK_ARG(p.img && p.wpk && p.bias && p.pos && p.x && p.counter && p.nseq > 0, "embed32: null/empty argument");
    HIPT_CHECK_ARG(p.ntx == 16 && p.nty % 8 == 0 && p.ntok == p.nty * p.ntx + 1, "embed32: token grid %dx%d / %d tokens", p.nty, p.ntx, p.ntok);
    HIPT_CHECK_ARG(((uintptr_t)p.img % 16) == 0 && p.im.row_stride % 4 == 0 && p.im.chan_stride % 4 == 0 && p.im.batch_stride % 4 == 0 && p.im.patch_w % 16 == 0,
                   "embed32: 16-byte aligned pixel rows required");
    const bool lnout = p.xn_out != nullptr;
    HIPT_CHECK_ARG(!lnout || (p.ln_w && p.ln_b && ((int64_t)p.nseq * p.ntok) % 16 == 0), "embed32: image output needs LayerNorm parameters and whole 16-row fragments");
    const int lds = 3 * UNIT + (lnout ? 3 : 1) * D * 4 + 16;
    static DevOnce once;
    HIPT_CUR_DEVICE(dev);
    if (!once.done[dev]) {
        if (hipFuncSetAttribute((const void*)embed32_kernel<0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void*)embed32_kernel<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void*)embed32_kernel<2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void*)embed32_kernel<0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void*)embed32_kernel<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void*)embed32_kernel<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            hipt_set_error("hipFuncSetAttribute(embed32) failed");
            return HIPT_E_LAUNCH;
        }
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) {
            hipt_set_error("embed32: cannot query the device");
            return HIPT_E_LAUNCH;
        }
        once.ncu[dev] = prop.multiProcessorCount;
        once.done[dev] = true;
    }
    const int ncu = once.ncu[dev];
    p.ntiles = p.nseq * (p.nty / 8);
    const int grid = p.ntiles < ncu ? p.ntiles : ncu;
    if (hipMemsetAsync(p.counter, 0, sizeof(int), st) != hipSuccess) {
        hipt_set_error("embed32: hipMemsetAsync(counter) failed");
        return HIPT_E_LAUNCH;
    }
    auto k = lnout ? (p.kind == 2 ? embed32_kernel<2, true> : (p.kind == 1 ? embed32_kernel<1, true> : embed32_kernel<0, true>))
                   : (p.kind == 2 ? embed32_kernel<2, false> : (p.kind == 1 ? embed32_kernel<1, false> : embed32_kernel<0, false>));
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, st, p);
    HIPT_CHECK_LAUNCH();
    return HIPT_OK;
}
