// Packed-weight layouts of the post-fusion U-Net shared by csrc/unet.hip (fp32 tensors) and csrc/unet_half.hip (bf16 tensors).
#pragma once
#include <cstdint>
#include <hip/hip_runtime.h>

namespace s2l {

// ---- packed layout ---------------------------------------------------------------------------------
struct ConvSpec {
  int cin, cout;
};
constexpr ConvSpec kUnetConvs[10] = {{3, 64},    {64, 64},   {64, 128}, {128, 128}, {128, 128},
                                     {128, 128}, {256, 128}, {128, 64}, {128, 64},  {64, 64}};
constexpr int kChunkFloats = 9 * 4 * 64 * 4;   // one (cout tile of 64, cin chunk of 16): 9 taps x 4 M-blocks x 64 lanes x 4

__host__ __device__ constexpr int64_t unet_w_off(int layer) {
  int64_t off = 0;
  for (int l = 0; l < layer; ++l)
    off += l == 0 ? 64 * 27 : (int64_t)(kUnetConvs[l].cout / 64) * (kUnetConvs[l].cin / 16) * kChunkFloats;
  return off;
}
constexpr int64_t kUnetBiasOff = unet_w_off(10);                   // folded biases, layer by layer
__host__ __device__ constexpr int64_t unet_b_off(int layer) {
  int64_t off = kUnetBiasOff;
  for (int l = 0; l < layer; ++l) off += kUnetConvs[l].cout;
  return off;
}
constexpr int64_t kUnetOutW = unet_b_off(10);                      // outc weight [3][64]
constexpr int64_t kUnetOutB = kUnetOutW + 192;                     // outc bias [4]
// transposed chunks of layers 1..9 for the input-gradient convolutions (s2l_unet_backward): the same chunk format with the
// roles of cin / cout swapped and the taps mirrored, dx = conv3x3(dz, W^T flipped)
constexpr int64_t kUnetWT = kUnetOutB + 4;
__host__ __device__ constexpr int64_t unet_wT_off(int layer) {
  int64_t off = kUnetWT;
  for (int l = 1; l < layer; ++l) off += (int64_t)(kUnetConvs[l].cin / 64) * (kUnetConvs[l].cout / 16) * kChunkFloats;
  return off;
}
constexpr int64_t kUnetPackedFloats = unet_wT_off(10);

// bf16 operand form of the 3x3 layers 1..9: a chunk = (64 output channels, 32 input channels): 9 taps x 2 k-steps x 2 M-blocks x
// 64 lanes x 8 bf16; lane l of an A quad holds W[row 32 mb + (l & 31)][k 16 s + 8 (l >> 5) + j] (unet_pack_conv16, csrc/unet.hip)
constexpr int kChunk16Halves = 9 * 2 * 2 * 64 * 8;
__host__ __device__ constexpr int64_t unet_w16_off(int layer) {
  int64_t off = 0;
  for (int l = 1; l < layer; ++l) off += (int64_t)(kUnetConvs[l].cout / 64) * (kUnetConvs[l].cin / 32) * kChunk16Halves;
  return off;
}
__host__ __device__ constexpr int64_t unet_wT16_off(int layer) {
  int64_t off = unet_w16_off(10);
  for (int l = 1; l < layer; ++l) off += (int64_t)(kUnetConvs[l].cin / 64) * (kUnetConvs[l].cout / 32) * kChunk16Halves;
  return off;
}
constexpr int64_t kUnetPacked16Halves = unet_wT16_off(10);

constexpr int kLvl[10] = {0, 0, 1, 1, 2, 2, 1, 1, 0, 0};      // resolution level of each convolution's output (0: H, 1: H/2, 2: H/4)

}  // namespace s2l
