// Layout of the bf16 training blob and of the bf16 activation / gradient tiles (BASELINE config 5 names bf16 for the
// training step).  Built by train_bf16.hip:pack, read by the bf16 forward / backward / weight-gradient kernels.
// All offsets are in bf16 elements ("halves").
//
// MFMA: v_mfma_f32_32x32x16_bf16.  A: lane l holds A[row l&31][k = 8*(l>>5) + j], j = 0..7 (one ds_read_b128);
// B: lane l holds B[k = 8*(l>>5) + j][col l&31]; D: reg r of lane l is D[(r&3) + 8*(r>>2) + 4*(l>>5)][l&31]
// (checked on the device by tools/ubench/mfma_bf16_layout.hip).  With D[feature][row of the batch], a lane's
// registers 8s..8s+7 of output block R are, after conversion to bf16, exactly its B operand for k-step 2R+s of the
// next layer if that layer's weights are packed with the K permutation
//     kfeat16(t, hh, j) = 32*(t>>1) + 8*(2*(t&1) + (j>>2)) + 4*hh + (j&3)
// so activations and gradients never leave registers between layers.
#pragma once
#include <cstdint>

namespace s2l {
namespace b16 {

__host__ __device__ constexpr int kfeat16(int t, int hh, int j) { return 32 * (t >> 1) + 8 * (2 * (t & 1) + (j >> 2)) + 4 * hh + (j & 3); }

constexpr int kSlabH = 32 * 256;   // 32 output rows x K=256: A image [t 16][lane 64][8]       (16 KiB)
constexpr int kSlabX = 32 * 128;   // 32 output rows x K=128 (embedding rows x): [t 8][lane 64][8] (8 KiB)
// forward stage s = 4*layer + quarter: the weights of output blocks R = 2q, 2q+1 of that layer:
//   [X slab R0][X slab R1][H slab R0][H slab R1]; X = folded G0 (layer 0) / G5 (layer 5), plain K order 16t + 8hh + j;
//   H = pts_linears[layer] (layer 5: columns 256..511), K order kfeat16; unused parts are zero and are never loaded.
//   The X part of stage 31 holds the output layer (one H-format slab, rows >= 3 zero).
constexpr int kStageF = 2 * kSlabX + 2 * kSlabH;        // 24576 halves = 48 KiB
constexpr int kFwdStages = 32;
constexpr int64_t OFF_FWD = 0;
// backward: U0 = output_linear^T (K = 16, 3 used): 8 slabs [lane 64][8]; then 30 stages of two H slabs in consumption
// order: W_7^T (4 stages), W_6^T (4), the audio columns of G5 transposed (1: rows = 64 audio dims), W_5b^T, W_4^T, W_3^T,
// W_2^T, W_1^T (4 each), the audio columns of G0 transposed (1).  W_l^T: rows = input feature, K = output feature in
// kfeat16 order.
constexpr int kSlabU0 = 64 * 8;
constexpr int kStageB = 2 * kSlabH;                      // 16384 halves = 32 KiB
constexpr int kBwdStages = 30;
constexpr int64_t OFF_BWD_U0 = OFF_FWD + int64_t(kFwdStages) * kStageF;
constexpr int64_t OFF_BWD_H = OFF_BWD_U0 + 8 * kSlabU0;
constexpr int64_t PACKED_HALVES = OFF_BWD_H + int64_t(kBwdStages) * kStageB;
// stage u -> (layer l whose W_l^T it holds, quarter), or the audio stages
__host__ __device__ constexpr bool bwd_stage_is_audio(int u) { return u == 8 || u == 29; }
__host__ __device__ constexpr int bwd_stage_layer(int u) { return u < 8 ? 7 - (u >> 2) : 5 - ((u - 9) >> 2); }
__host__ __device__ constexpr int bwd_stage_quarter(int u) { return u < 8 ? (u & 3) : ((u - 9) & 3); }

// Saved activations / gradients ("images"): exactly what a lane holds after a 32-feature block's epilogue, so the forward
// and backward kernels store them with two 16-byte stores per lane and no transposition:
//     image[row group of 32][block R = F/32][half 2][lane = n + 32 hh][8 bf16],  element 4 (a & 1) + c of half a >> 1  =
// feature 32R + 8a + 4hh + c of row n.  The two 16-byte pieces of a lane sit 1 KiB apart, so that each of the two store
// instructions of a block writes 1 KiB of CONTIGUOUS memory: 16 full 64-byte sectors.  (Until late in round 2 a lane's 32 bytes
// were adjacent; every store instruction then touched 32 half-filled sectors, and tools/ubench/gen_mfma_shadow.py's
// gstore_*_pitch32 / pitch16 show that the texture path takes twice as long for that: 82 against 40 cycles per instruction
// and CU.)  For a fixed hh and both halves this is a row-major [32 rows][16 features] bf16 matrix whose 8-byte groups are 4
// consecutive features -- rebuilt with 32-byte rows in LDS by the weight-gradient kernel's copy, it is the shape
// ds_read_b64_tr_b16 turns into "8 consecutive rows of one feature per lane", which is what BOTH operands of the
// weight-gradient GEMM dW = dz^T h need (the reduction runs over rows).
constexpr int kGroupRows = 32;
constexpr int kWgRows = 256;       // rows per workgroup tile of the forward / backward kernels
// offset (in halves) of block R of a row group (2 KiB = 1024 halves per block); of the 16-byte piece `half` of a lane in it;
// of the 8-byte group a (features 8a + 4hh ..) of a lane in it
__host__ __device__ constexpr int64_t image_off(int64_t group, int n_blocks, int R) { return (group * n_blocks + R) * 1024; }
__host__ __device__ constexpr int image_piece(int half, int lane) { return (half * 64 + lane) * 8; }
__host__ __device__ constexpr int image_quad(int a, int lane) { return image_piece(a >> 1, lane) + 4 * (a & 1); }
// ReLU masks: uint32 [layer 8][group of 32 rows][stage q 4][lane n + 32 hh]: the lane's 32 values of the stage (blocks 2q, 2q + 1)
// are 16 bf16 pairs d = 8 which + 2 a + p; bit 15 - d = (low half of pair d is non-zero), bit 31 - d = (high half is)

}  // namespace b16
}  // namespace s2l
