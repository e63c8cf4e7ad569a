// s2l_rgb_forward: the exact drop-in for TalkingFace.rgb_forward (tf_nerf.py:225-285) on arbitrary
// [N,66] rows: embed the rows (frontend.hip) and run the general-row MLP on the LDS-DMA weight ring
// (rows.hip).  The clip renderer (s2l_render_lip, the benchmarked kernel) lives in render.hip.
#include "s2l_common.h"

namespace s2l {

int launch_embed_rows(const float* packed, const float* uv_audio, int64_t time_index, float* x, int64_t n_rows,
                      hipStream_t st);
int launch_rows_fwd(const float* packed, const float* x, float* out, float* hsave, int64_t n_rows, hipStream_t st);
int launch_rows_bwd(const float* packed, const float* drgb, const float* hsave, float* dzsave, float* dxa, int64_t n_rows,
                    hipStream_t st);

int launch_general_mlp(const float* packed, const float* x, float* out, float* hsave, int64_t n_rows, hipStream_t st) {
  return launch_rows_fwd(packed, x, out, hsave, n_rows, st);
}

int launch_general_mlp_bwd(const float* packed, const float* drgb, const float* hsave, float* dzsave, float* dxa,
                           int64_t n_rows, hipStream_t st) {
  return launch_rows_bwd(packed, drgb, hsave, dzsave, dxa, n_rows, st);
}

}  // namespace s2l

extern "C" int s2l_rgb_forward(const float* packed, const float* uv_audio, int64_t time_index, float* xbuf, float* out,
                               int64_t n_rows, s2l_stream_t stream) {
  if (n_rows < 0) return S2L_E_SIZE;
  if (n_rows == 0) return S2L_OK;
  if (!packed || !uv_audio || !xbuf || !out) return S2L_E_NULL;
  if (s2l::misaligned16(packed) || s2l::misaligned16(xbuf)) return S2L_E_ALIGN;
  hipStream_t st = static_cast<hipStream_t>(stream);
  int rc = s2l::launch_embed_rows(packed, uv_audio, time_index, xbuf, n_rows, st);
  if (rc) return rc;
  return s2l::launch_general_mlp(packed, xbuf, out, nullptr, n_rows, st);
}
