// Layout of the packed weight blob (device, fp32).  Offsets are in floats; every section starts
// on a 16-byte boundary.  Built by pack.hip, read by every kernel.
#pragma once
#include <cstdint>

namespace s2l {

constexpr int kW = 256;          // hidden width
constexpr int kEmb = 42;         // Embedder(10, 2) output dims
constexpr int kAud = 64;         // audio feature dims
constexpr int kTime = 20;        // PositionalEncodingTime dims
constexpr int kGenK = 128;       // general-mode first-layer K: 42 + 64 + 20, zero-padded to 128
constexpr int kHidden = 7;       // MFMA layers: pts_linears 1,2,3,4,5[:,256:],6,7
constexpr int kSlab = 16 * 64 * 4;   // one M-block (16 output features) of a K=256 layer, floats

// A-operand order of one M-block slab (v_mfma_f32_16x16x4_f32, K=256):
//   slab[(j4*64 + lane)*4 + jj] = W[mb*16 + (lane&15)][kfeat(j4*4+jj, lane>>4)]
//   kfeat(j, q) = (j>>2)*16 + 4*q + (j&3)
// so the k-step-j B operand of lane (q, pixel) is exactly accumulator register (j&3) of the
// previous layer's M-block (j>>2): activations stay in registers between layers.
__host__ __device__ constexpr int kfeat(int j, int q) { return (j >> 2) * 16 + 4 * q + (j & 3); }
// General-mode input rows x[128] use kin(j, q) = 32*q + j (each lane reads 8 contiguous quads).

constexpr int64_t align4(int64_t x) { return (x + 3) & ~int64_t(3); }

constexpr int64_t OFF_WMLP = 0;                                    // [7][16 mb][16 j4][64][4]
constexpr int64_t OFF_WOUT = OFF_WMLP + int64_t(kHidden) * 16 * kSlab;   // [16 j4][64][4], rows>=3 zero
constexpr int64_t OFF_WG0 = OFF_WOUT + kSlab;                      // folded W0 [Wuv|Wa|Wt|0]: [16][8][64][4]
constexpr int64_t OFF_WG5 = OFF_WG0 + 16 * (kSlab / 2);            // folded W5a[Wuv'|Wa'|Wt'|0]
constexpr int64_t OFF_BIAS = OFF_WG5 + 16 * (kSlab / 2);           // [7][256] biases of pts 1..7
constexpr int64_t OFF_BOUT = OFF_BIAS + kHidden * kW;              // [4]
constexpr int64_t OFF_BG0 = OFF_BOUT + 4;                          // [256] W0 (buv+ba+bt) + b0
constexpr int64_t OFF_BG5 = OFF_BG0 + kW;                          // [256] W5a(buv'+ba'+bt') + b5
constexpr int64_t OFF_W0T = OFF_BG5 + kW;                          // pts0^T       [k 256][n 256]
constexpr int64_t OFF_W5AT = OFF_W0T + kW * kW;                    // pts5[:, :256]^T
constexpr int64_t OFF_WUVT = OFF_W5AT + kW * kW;                   // fc_uv^T      [42][256]
constexpr int64_t OFF_WUVST = OFF_WUVT + kEmb * kW;                // fc_uv_skip^T
constexpr int64_t OFF_WAT = OFF_WUVST + kEmb * kW;                 // fc_audio^T   [64][256]
constexpr int64_t OFF_WAST = OFF_WAT + kAud * kW;
constexpr int64_t OFF_WTT = OFF_WAST + kAud * kW;                  // fc_time^T    [20][256]
constexpr int64_t OFF_WTST = OFF_WTT + kTime * kW;
constexpr int64_t OFF_BSUM0 = OFF_WTST + kTime * kW;               // buv + ba + bt
constexpr int64_t OFF_BSUM5 = OFF_BSUM0 + kW;                      // skip biases summed
constexpr int64_t OFF_B0 = OFF_BSUM5 + kW;                         // pts0 bias
constexpr int64_t OFF_B5 = OFF_B0 + kW;                            // pts5 bias
constexpr int64_t OFF_DIV = OFF_B5 + kW;                           // [16] div_term (10 used)
// audio encoder, conv weights transposed to [cin][k][cout], fc to [in][out]
constexpr int64_t OFF_C0W = OFF_DIV + 16;                          // [29][3][32]
constexpr int64_t OFF_C0B = OFF_C0W + align4(29 * 3 * 32);
constexpr int64_t OFF_C2W = OFF_C0B + 32;                          // [32][3][32]
constexpr int64_t OFF_C2B = OFF_C2W + 32 * 3 * 32;
constexpr int64_t OFF_C4W = OFF_C2B + 32;                          // [32][3][64]
constexpr int64_t OFF_C4B = OFF_C4W + 32 * 3 * 64;
constexpr int64_t OFF_C6W = OFF_C4B + 64;                          // [64][3][64]
constexpr int64_t OFF_C6B = OFF_C6W + 64 * 3 * 64;
constexpr int64_t OFF_F0W = OFF_C6B + 64;                          // [64][64]
constexpr int64_t OFF_F0B = OFF_F0W + 64 * 64;
constexpr int64_t OFF_F2W = OFF_F0B + 64;
constexpr int64_t OFF_F2B = OFF_F2W + 64 * 64;
// ---- training (backward) sections -------------------------------------------------------------
// plain folded first/skip matrices G0 = W0 [Wuv|Wa|Wt|0], G5 = W5a [Wuv'|Wa'|Wt'|0]: [256][128]
constexpr int64_t OFF_G0 = OFF_F2B + 64;
constexpr int64_t OFF_G5 = OFF_G0 + kW * kGenK;
// transposed hidden layers in A-operand order for dgrad: slab(L, mb)[(j4*64+lane)*4+jj] =
//   W_L[kfeat(j4*4+jj, lane>>4)][mb*16 + (lane&15)]   (dh_in[mb-block] = W^T dz, k runs over OUT features)
constexpr int64_t OFF_WMLPT = OFF_G5 + kW * kGenK;                  // [7][16][16][64][4]
// output layer transposed, K = 4 (3 used): [16 mb][64 lanes] one float per lane: Wout[lane>>4][mb*16+(lane&15)]
constexpr int64_t OFF_WOUTT = OFF_WMLPT + int64_t(kHidden) * 16 * kSlab;
// audio columns (42..105) of G0 / G5 transposed: d a = G[:, 42:106]^T dz: [4 mb][16 j4][64][4]
constexpr int64_t OFF_G0AT = OFF_WOUTT + 16 * 64;
constexpr int64_t OFF_G5AT = OFF_G0AT + 4 * kSlab;
constexpr int64_t PACKED_FLOATS = OFF_G5AT + 4 * kSlab;

static_assert(OFF_WOUT % 4 == 0 && OFF_BIAS % 4 == 0 && OFF_W0T % 4 == 0 && OFF_C0B % 4 == 0, "16B sections");

}  // namespace s2l
