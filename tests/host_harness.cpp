// TEST INFRASTRUCTURE ONLY: compiles ld_amd/csrc/ld_math.h (the scalar math the
// gfx950 kernels are built from) for the host with g++, so the CPU test-suite
// can check those formulas against the golden vectors without a GPU.  The
// product never links or loads this.
#include "../ld_amd/csrc/ld_math.h"

extern "C" {
float h_iou(const float* a, const float* b) {
  return ld::iou_pair(ld::Box{a[0], a[1], a[2], a[3]}, ld::Box{b[0], b[1], b[2], b[3]});
}
float h_diou(const float* a, const float* b) {
  return ld::diou_pair(ld::Box{a[0], a[1], a[2], a[3]}, ld::Box{b[0], b[1], b[2], b[3]});
}
float h_dist(float ax, float ay, float gx, float gy) {
  return ld::centre_dist(ax, ay, gx, gy);
}
float h_giou(const float* p, const float* t, float eps, float* iou, float* g) {
  return ld::giou_loss_grad(ld::Box{p[0], p[1], p[2], p[3]},
                            ld::Box{t[0], t[1], t[2], t[3]}, eps, iou, g);
}
float h_kl17(const float* s, const float* t, float T, float* d) {
  return ld::kl_rows<17>(s, t, 1.0f / T, T, d);
}
float h_expect17(const float* s, float* p) { return ld::softmax_expect<17>(s, p); }
float h_dfl17(const float* s, float y, float* wl, float* wr, int* yl) {
  float p[17];
  ld::softmax_expect<17>(s, p);
  return ld::dfl_side<17>(s, p, y, wl, wr, yl);
}
float h_qfl_neg(float x, float* dq) { return ld::qfl_neg(x, dq); }
float h_qfl_pos(float x, float s, float* dq) { return ld::qfl_pos(x, s, dq); }
float h_clamp_dist(float d, float rm) { return ld::clamp_dist(d, rm); }
void h_anchor(int x, int y, int stride, float half, float* out) {
  ld::Box b = ld::anchor_box(x, y, stride, half);
  out[0] = b.x1; out[1] = b.y1; out[2] = b.x2; out[3] = b.y2;
}
}
