// The resident refinement loop behind one C entry point (`dim_refine_4iter`): what deepim/core/tester.py `Refiner._loop`
// + deepim/symbols/deepIM_flownet.py `FlowNetHip.forward_test` enqueue from Python, for a host that has no torch.
//
// Replaces the inner loop of pred_eval (/root/reference/deepim/core/tester.py:476-598) for a batch of B pairs that stays in HBM:
//   ZoomMask + ZoomImageWithFactor + Concat -> FlowNetS encoder (Winograd / direct MFMA layers) -> fc6 -> fc7 / rot / trans ->
//   RT_transform -> render -> box_rendered mask -> next iteration
// for the shipped FAST_TEST test graph (8-channel input, UPDATE_MASK 'box_rendered').  create() packs the weights and allocates every
// buffer; run() only enqueues kernels on the caller's stream (no allocation, no synchronisation: it can be captured into a hipGraph);
// the layer plans (tile, split-K, Winograd tile) are the ones the Python executor uses, so both drive the same launches and produce
// the same bits (tests/test_gpu_refiner_capi.py).
#include <cstring>
#include <string>
#include <vector>

#include "common.h"

namespace dim {

struct Layer {
  const char* name;
  int cout, k, s, p;
};
static const Layer kEncoder[10] = {{"flow_conv1", 64, 7, 2, 3}, {"conv2", 128, 5, 2, 2},   {"conv3", 256, 5, 2, 2},  {"conv3_1", 256, 3, 1, 1},
                                   {"conv4", 512, 3, 2, 1},     {"conv4_1", 512, 3, 1, 1}, {"conv5", 512, 3, 2, 1},  {"conv5_1", 512, 3, 1, 1},
                                   {"conv6", 1024, 3, 2, 1},    {"conv6_1", 1024, 3, 1, 1}};

// (the launch plans come from dim_conv_auto_plan / dim_winograd_gemm_tile in conv.hip: one copy for this file and for FlowNetHip)

struct LayerPlan {
  int kind;  // 0 direct, 1 Winograd F(4x4,3x3), 2 phase-image Winograd (5x5 / stride 2), 3 phase-image minimal filtering (3x3 / stride 2)
  int h, w, cin, ho, wo;
  int tile, splits;
  float* w_packed;
  float* bias;
  float* out;
};

}  // namespace dim

using namespace dim;

struct dim_refiner {
  dim_refiner_desc d;
  std::vector<void*> allocs;
  LayerPlan L[10];
  float *fc6_w, *fc6_b, *fc7_w, *fc7_b, *rot_w, *rot_b, *trans_w, *trans_b;
  float *X, *workspace, *fc6, *zoom_factor, *image_rendered, *mask_observed, *mask_rendered;
  int *bbox_obs, *bbox_ren, *bbox_ras, *bbox_ras2, *bbox_box;
  void* raster_ws;
  float plane_means[3];
};

namespace {

int dev_alloc(dim_refiner* r, void** p, size_t bytes) {
  hipError_t e = hipMalloc(p, bytes ? bytes : 16);
  if (e != hipSuccess) return set_err(DIM_ERR_LAUNCH, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e));
  r->allocs.push_back(*p);
  return DIM_OK;
}

const float* find_param(const char* const* names, const float* const* ptrs, int n, const std::string& want) {
  for (int i = 0; i < n; ++i)
    if (want == names[i]) return ptrs[i];
  return nullptr;
}

}  // namespace

extern "C" {

int dim_refiner_destroy(dim_refiner* r) {
  if (!r) return DIM_OK;
  for (void* p : r->allocs) (void)hipFree(p);
  delete r;
  return DIM_OK;
}

int dim_refiner_create(dim_refiner** out, const dim_refiner_desc* desc, const char* const* param_names, const float* const* param_ptrs,
                       int n_params, void* stream) {
  DIM_REQUIRE(out && desc && param_names && param_ptrs, "null pointer");
  const dim_refiner_desc& d = *desc;
  DIM_REQUIRE(d.B > 0 && d.H == 480 && d.W == 640 && d.test_iter >= 1, "B > 0, H x W = 480 x 640, test_iter >= 1 required");
  DIM_REQUIRE(d.verts && d.uvs && d.faces && d.mesh_table && d.textures && d.tex_table && d.n_classes > 0 && d.vmax > 0 && d.fmax > 0,
              "mesh table missing");
  dim_refiner* r = new dim_refiner();
  r->d = d;
  for (int c = 0; c < 3; ++c) r->plane_means[c] = d.pixel_means_bgr[2 - c];  // blob plane c holds BGR channel 2 - c
  const int B = d.B;
  int rc = DIM_OK;
#define TRY(x)                 \
  if ((rc = (x)) != DIM_OK) {  \
    dim_refiner_destroy(r);    \
    return rc;                 \
  }
  auto need = [&](const std::string& n) -> const float* { return find_param(param_names, param_ptrs, n_params, n); };
  // ---- encoder: plans, packed weights, activations
  long max_ws = 4;
  int h = d.H, w = d.W, c = 8;
  for (int i = 0; i < 10; ++i) {
    const Layer& ly = kEncoder[i];
    LayerPlan& P = r->L[i];
    const float* wsrc = need(std::string(ly.name) + "_weight");
    const float* bsrc = need(std::string(ly.name) + "_bias");
    if (!wsrc || !bsrc) {
      dim_refiner_destroy(r);
      return set_err(DIM_ERR_ARG, "parameter %s_weight / %s_bias missing", ly.name, ly.name);
    }
    P.h = h; P.w = w; P.cin = c;
    P.ho = (h + 2 * ly.p - ly.k) / ly.s + 1;
    P.wo = (w + 2 * ly.p - ly.k) / ly.s + 1;
    P.bias = const_cast<float*>(bsrc);  // biases are read where the caller keeps them (they must outlive the refiner)
    const long M = (long)B * P.ho * P.wo;
    TRY(dev_alloc(r, (void**)&P.out, (size_t)M * ly.cout * 4));
    if (ly.k == 3 && ly.s == 1 && ly.p == 1) {
      P.kind = 1;
      P.tile = dim_winograd_gemm_tile(ly.cout, (long)B * ((h + 3) / 4) * ((w + 3) / 4));
      P.splits = 1;
      TRY(dev_alloc(r, (void**)&P.w_packed, (size_t)dim_winograd_packed_weight_floats(ly.cout, c, 4) * 4));
      TRY(dim_winograd_pack_weight(wsrc, P.w_packed, ly.cout, c, 4, stream));
      max_ws = std::max(max_ws, dim_winograd_workspace_floats(B, h, w, c, ly.cout, 4));
    } else if (ly.k == 5 && ly.s == 2 && ly.p == 2) {
      P.kind = 2;
      P.tile = dim_winograd_gemm_tile(ly.cout, (long)B * ((P.ho + 3) / 4) * ((P.wo + 3) / 4));
      P.splits = 1;
      TRY(dev_alloc(r, (void**)&P.w_packed, (size_t)dim_winograd5x5s2_packed_weight_floats(ly.cout, c) * 4));
      TRY(dim_winograd5x5s2_pack_weight(wsrc, P.w_packed, ly.cout, c, stream));
      max_ws = std::max(max_ws, dim_winograd5x5s2_workspace_floats(B, h, w, c, ly.cout));
    } else if (ly.k == 3 && ly.s == 2 && ly.p == 1 && dim_winograd3x3s2_use(h, w, c, ly.cout)) {
      P.kind = 3;
      P.tile = dim_winograd_gemm_tile_planes(ly.cout, (long)B * ((P.ho + 3) / 4) * ((P.wo + 3) / 4), 81);
      P.splits = 1;
      TRY(dev_alloc(r, (void**)&P.w_packed, (size_t)dim_winograd3x3s2_packed_weight_floats(ly.cout, c) * 4));
      TRY(dim_winograd3x3s2_pack_weight(wsrc, P.w_packed, ly.cout, c, stream));
      max_ws = std::max(max_ws, dim_winograd3x3s2_workspace_floats(B, h, w, c, ly.cout));
    } else {
      P.kind = 0;
      const int nchunks = c == 8 ? (ly.k * ly.k + 3) / 4 : ly.k * ly.k * (c / 32);
      TRY(dim_conv_auto_plan(M, ly.cout, nchunks, c, &P.tile, &P.splits));
      if (c == 8 && ly.k == 7 && ly.s == 2 && ly.cout == 64) { P.tile = 6; P.splits = 1; }  // LDS-halo first-layer kernel, as FlowNetHip
      TRY(dev_alloc(r, (void**)&P.w_packed, (size_t)dim_conv2d_packed_weight_floats(ly.cout, c, ly.k, ly.k) * 4));
      TRY(dim_conv2d_pack_weight(wsrc, P.w_packed, ly.cout, c, ly.k, ly.k, stream));
      if (P.splits != 1) max_ws = std::max(max_ws, dim_conv2d_workspace_floats(B, h, w, c, ly.cout, ly.k, ly.k, ly.s, ly.p, P.splits));
    }
    h = P.ho; w = P.wo; c = ly.cout;
  }
  if (!(h == 8 && w == 10 && c == 1024)) {
    dim_refiner_destroy(r);
    return set_err(DIM_ERR_ARG, "encoder geometry");
  }
  // ---- heads
  const float* fc6w = need("fc6_weight");
  r->fc6_b = const_cast<float*>(need("fc6_bias"));
  r->fc7_w = const_cast<float*>(need("fc7_weight")); r->fc7_b = const_cast<float*>(need("fc7_bias"));
  r->rot_w = const_cast<float*>(need("rot_weight")); r->rot_b = const_cast<float*>(need("rot_bias"));
  r->trans_w = const_cast<float*>(need("trans_weight")); r->trans_b = const_cast<float*>(need("trans_bias"));
  if (!fc6w || !r->fc6_b || !r->fc7_w || !r->fc7_b || !r->rot_w || !r->rot_b || !r->trans_w || !r->trans_b) {
    dim_refiner_destroy(r);
    return set_err(DIM_ERR_ARG, "fc6 / fc7 / rot / trans parameters missing");
  }
  TRY(dev_alloc(r, (void**)&r->fc6_w, (size_t)256 * 81920 * 4));
  TRY(dim_fc_pack_weight(fc6w, r->fc6_w, 256, 1024, 8, 10, stream));
  max_ws = std::max(max_ws, dim_fc_fwd_workspace_floats(1024, 8, 10, 256));
  // ---- buffers
  const size_t plane = (size_t)d.H * d.W * 4;
  TRY(dev_alloc(r, (void**)&r->X, (size_t)B * plane * 8));
  TRY(dev_alloc(r, (void**)&r->workspace, (size_t)max_ws * 4));
  TRY(dev_alloc(r, (void**)&r->fc6, (size_t)B * 256 * 4));
  TRY(dev_alloc(r, (void**)&r->zoom_factor, (size_t)B * 4 * 4));
  TRY(dev_alloc(r, (void**)&r->image_rendered, (size_t)B * plane * 3));
  TRY(dev_alloc(r, (void**)&r->mask_observed, (size_t)B * plane));
  TRY(dev_alloc(r, (void**)&r->mask_rendered, (size_t)B * plane));
  TRY(dev_alloc(r, (void**)&r->bbox_obs, (size_t)B * 16));
  TRY(dev_alloc(r, (void**)&r->bbox_ren, (size_t)B * 16));
  TRY(dev_alloc(r, (void**)&r->bbox_ras, (size_t)B * 16));
  TRY(dev_alloc(r, (void**)&r->bbox_ras2, (size_t)B * 16));   // the boxes of two consecutive renders: one is the next one's dirty-box hint
  TRY(dev_alloc(r, (void**)&r->bbox_box, (size_t)B * 16));
  TRY(dev_alloc(r, &r->raster_ws, (size_t)dim_raster_workspace_bytes(B, d.vmax, d.H, d.W)));
  {   // the rasteriser's header must not hold what a previous owner of this memory left there (deepim_hip.h, rasteriser section)
    hipError_t e = hipMemsetAsync(r->raster_ws, 0, 256, as_stream(stream));
    if (e != hipSuccess) { dim_refiner_destroy(r); return set_err(DIM_ERR_LAUNCH, "hipMemsetAsync: %s", hipGetErrorString(e)); }
  }
#undef TRY
  *out = r;
  return DIM_OK;
}

// poses_iter (T,B,3,4), se3_iter (T,B,7), status_iter (T,B): outputs, device.  The six input blobs are read, never written.
int dim_refiner_run(dim_refiner* r, const float* image_observed, const float* image_rendered, const float* mask_observed,
                    const float* mask_rendered, const float* src_pose, const int* class_index, float* poses_iter, float* se3_iter,
                    int* status_iter, void* stream) {
  DIM_REQUIRE(r && image_observed && image_rendered && mask_observed && mask_rendered && src_pose && class_index && poses_iter && se3_iter &&
                  status_iter,
              "null pointer");
  const dim_refiner_desc& d = r->d;
  const int B = d.B, H = d.H, W = d.W, T = d.test_iter;
  int rc;
#define TRY(x) \
  if ((rc = (x)) != DIM_OK) return rc;
  const float* img_ren = image_rendered;
  const float* m_obs = mask_observed;
  const float* m_ren = mask_rendered;
  const float* pose = src_pose;
  const int* bb_obs = nullptr;
  const int* bb_ren = nullptr;
  for (int it = 0; it < T; ++it) {
    float* se3 = se3_iter + (long)it * B * 7;
    float* pose_out = poses_iter + (long)it * B * 12;
    int* status = status_iter + (long)it * B;
    // ---- ZoomMask + ZoomImageWithFactor + Concat (deepIM_flownet.py:783-806, :53-60)
    if (!bb_obs) {
      TRY(dim_mask_bbox(m_obs, B, 1, H, W, 0, 0.3f, nullptr, r->bbox_obs, stream));
      bb_obs = r->bbox_obs;
    }
    if (!bb_ren) {
      TRY(dim_mask_bbox(m_ren, B, 1, H, W, 0, 0.2f, nullptr, r->bbox_ren, stream));
      bb_ren = r->bbox_ren;
    }
    TRY(dim_zoom_factor(bb_obs, bb_ren, pose, d.K9, B, H, W, r->zoom_factor, status, stream));
    TRY(dim_zoom_net_input(image_observed, img_ren, m_obs, m_ren, r->zoom_factor, r->X, B, H, W, r->plane_means, nullptr, nullptr, nullptr,
                           nullptr, stream));
    // ---- encoder (:67-198)
    const float* x = r->X;
    for (int i = 0; i < 10; ++i) {
      const Layer& ly = kEncoder[i];
      const LayerPlan& P = r->L[i];
      if (P.kind == 1) {
        TRY(dim_conv2d_fwd_winograd(x, P.w_packed, P.bias, P.out, r->workspace, B, P.h, P.w, P.cin, P.cin, ly.cout, ly.cout, 0, 0.1f, P.tile, 4,
                                    nullptr, stream));
      } else if (P.kind == 2) {
        TRY(dim_conv2d_fwd_winograd5x5s2(x, P.w_packed, P.bias, P.out, r->workspace, B, P.h, P.w, P.cin, P.cin, ly.cout, ly.cout, 0, 0.1f,
                                         P.tile, nullptr, stream));
      } else if (P.kind == 3) {
        TRY(dim_conv2d_fwd_winograd3x3s2(x, P.w_packed, P.bias, P.out, r->workspace, B, P.h, P.w, P.cin, P.cin, ly.cout, ly.cout, 0, 0.1f,
                                         P.tile, nullptr, stream));
      } else {
        TRY(dim_conv2d_fwd(x, P.w_packed, P.bias, P.out, r->workspace, B, P.h, P.w, P.cin, ly.cout, ly.k, ly.k, ly.s, ly.p, 0.1f, P.splits,
                           P.tile, stream));
      }
      x = P.out;
    }
    TRY(dim_fc_fwd(x, r->fc6_w, r->fc6_b, r->fc6, r->workspace, B, 1024, 8, 10, 256, 0.1f, stream));
    // ---- fc7, rot, trans, inverse ZoomTrans -> se3 (:203-208, :956-971); RT_transform (tester.py:525-532)
    TRY(dim_pose_head_fwd(r->fc6, r->fc7_w, r->fc7_b, r->rot_w, r->rot_b, r->trans_w, r->trans_b, r->zoom_factor, se3, nullptr, B, stream));
    TRY(dim_se3_compose(pose, se3, pose_out, nullptr, B, d.rot_coord, d.T_means, d.T_stds, stream));
    if (it < T - 1) {
      // ---- render + update_data_batch (tester.py:563-590, data_pair.py:103-114)
      // (from the second render on, the planes hold the previous render: background outside ITS box, which is not written again)
      int* bb_new = (it & 1) ? r->bbox_ras2 : r->bbox_ras;
      const int* bb_prev = it == 0 ? nullptr : ((it & 1) ? r->bbox_ras : r->bbox_ras2);
      TRY(dim_raster_render_dirty(d.verts, nullptr, d.uvs, d.faces, d.mesh_table, d.n_classes, d.vmax, d.fmax, d.textures, d.tex_table,
                                  class_index, pose_out, d.K9, B, H, W, d.znear, d.zfar, d.tex_bilinear, nullptr, nullptr, 0.f, r->plane_means,
                                  0.2f, r->raster_ws, r->image_rendered, nullptr, r->mask_rendered, nullptr, bb_new, status,
                                  d.znear > 0.2f ? bb_prev : nullptr, stream));
      TRY(dim_box_mask(bb_new, r->mask_observed, B, H, W, r->bbox_box, stream));
      img_ren = r->image_rendered;
      m_obs = r->mask_observed;
      m_ren = r->mask_rendered;
      bb_obs = r->bbox_box;
      bb_ren = bb_new;
      pose = pose_out;
    }
  }
#undef TRY
  return DIM_OK;
}

}  // extern "C"
