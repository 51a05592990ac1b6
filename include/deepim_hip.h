/* libdeepim_hip.so -- C ABI of the MI355X-native DeepIM refinement hot path.
 *
 * Drop-in boundary: these are the entry points a maintainer of wangg12/mx-DeepIM would bind
 * (ctypes) in place of
 *   - the numpy/MXNet bodies of the Python custom ops in deepim/operator_py/ (every .py there),
 *   - the one native function the reference has today,
 *       void _flow(float* flow, float* valid, float* depth_src, float* depth_tgt, float* KT,
 *                  float* Kinv, int batch_size, int height, int width, int device_id)
 *       (lib/flow_c/gpu_flow.hpp:1-3, Cython binding lib/flow_c/gpu_flow.pyx:24-41),
 *   - the glumpy renderer object lib/render_glumpy/render_py_multi.py:49-147,
 *   - MXNet's Convolution / FullyConnected operators used by deepim/symbols/deepIM_flownet.py.
 *
 * Conventions
 *   - every function returns DIM_OK (0) or a negative DIM_ERR_* code; dim_last_error() returns a
 *     thread-local description of the last failure on the calling thread.
 *   - pointers are DEVICE pointers (caller-owned, fp32, C-contiguous) unless the parameter name
 *     ends in a digit count (`K9`, `means3`, `T_means3`, ...): those are small HOST arrays.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  Calls only enqueue work;
 *     nothing here allocates, frees or synchronises, so every call is hipGraph-capturable.
 *   - image-like tensors are NCHW as in the reference's blobs; the conv stack's activations are
 *     NHWC (this library's internal layout, produced by dim_zoom_net_input).
 */
#ifndef DEEPIM_HIP_H_
#define DEEPIM_HIP_H_

#ifdef __cplusplus
extern "C" {
#endif

#define DIM_OK 0
#define DIM_ERR_ARG (-1)
#define DIM_ERR_LAUNCH (-2)
/* per-sample status words (device int32).  Contract: dim_zoom_factor OVERWRITES status[b] (0 or its two bits) -- it is the first
 * kernel of an iteration, so a replayed loop (dim_refiner_run, Refiner._loop) starts every iteration's row clean without a fill;
 * dim_raster_render* OR their bits into the word.  A caller of a standalone render zeroes the words first (dim_fill_words), and a
 * caller that wants both kinds of bits in one word calls dim_zoom_factor BEFORE the render, as the loop does. */
#define DIM_STATUS_OBS_BOX_EMPTY 1 /* dim_zoom_factor: observed box empty (the reference raises) */
#define DIM_STATUS_REN_BOX_EMPTY 2 /* dim_zoom_factor: rendered box empty */
#define DIM_STATUS_BAD_CLASS 4     /* dim_raster_render*: class_index outside [0, n_classes): sample rendered as background */
#define DIM_STATUS_BAD_FACE 8      /* dim_raster_render*: a z-buffer key named a face outside the mesh (pixel left black) */

const char* dim_last_error(void);
/* library / device probe: fills name (<= n bytes), returns number of compute units or <0 */
int dim_device_info(char* name, int n);

/* Device-to-device copy of `nwords` 4-byte words as a plain kernel launch.  The refinement loop uses it instead of
 * hipMemcpyAsync / hipMemsetAsync inside captured hipGraphs: a captured memset node was observed to overlap the kernel
 * after it on replay (ROCm 7.2), so no copy / fill node is left in the graph (deepim/core/tester.py Refiner._loop). */
int dim_copy_words(void* dst, const void* src, long nwords, void* stream);
/* The same for `rows` runs of `width_words` words with different pitches on the two sides: a channel window [:, a:b] of an
 * (O, I, kh, kw) weight is one run of (b - a) kh kw words per output channel (the training executor's first layer when the network
 * input has 6 or 10 channels: reference get_convs, deepim/symbols/deepIM_flownet.py:33-66). */
int dim_copy_rows(void* dst, long dst_pitch_words, const void* src, long src_pitch_words, long rows, long width_words, void* stream);
/* dst += src over the same geometry, as floats (the skip connections of the decoder add a channel range of a concat-gradient buffer to
 * an encoder gradient: get_convs Concat2 / Concat3, deepim/symbols/deepIM_flownet.py:236-299), and a constant fill (loss sums) -- so
 * that no vendor elementwise kernel sits on the training or test path. */
int dim_add_rows(float* dst, long dst_pitch, const float* src, long src_pitch, long rows, long width, void* stream);
int dim_fill_words(void* dst, long nwords, unsigned value, void* stream);

/* ---------------------------------------------------------------- zoom ops
 * bbox of {x > thr} (mode 0, C==1) or {sum_c (x_c + means3[c]) > thr} (mode 1, C==3):
 *   bbox[b] = {min_x, max_x, min_y, max_y}, empty = {W,-1,H,-1}.
 * replaces zoom_mask.py:36-70 / zoom_image.py:34-70 (np.max / np.nonzero on host copies). */
int dim_mask_bbox(const float* x, int B, int C, int H, int W, int mode, float thr, const float* means3, int* bbox, void* stream);

/* zoom window rule -> zoom_factor (B,4) = [wx, wy, tx, ty]   (zoom_mask.py:71-117).
 * status (B) may be NULL; bit0 = observed box empty (reference raises), bit1 = rendered box empty
 * (reference prints "NO POINT VALID IN MASK rendered" and uses the observed box). */
int dim_zoom_factor(const int* bbox_observed, const int* bbox_rendered, const float* src_pose, const float* K9, int B, int H, int W,
                    float* zoom_factor, int* status, void* stream);

/* affine bilinear gather of C planes with a zoom factor (GridGenerator + BilinearSampler fused).
 *   inverse   1 = inverse zoom (zoom_flow.py:35-44)
 *   pre       0 none | 1 binarise input at 0.2          (zoom_mask.py:39-46)
 *   post      0 none | 1 mx.nd.round | 2 round(x-0.45)   (zoom_mask.py:121-129, zoom_flow.py:74-76)
 *   add3      HOST per-plane constant added before / removed after sampling (pixel means), or NULL
 *   scale_mode 0 none | 1 divide by wx | 2 multiply by wx (zoom_flow.py:59-66) */
int dim_zoom_planes(const float* x, const float* zoom_factor, float* y, int B, int C, int H, int W, int inverse, int pre, int post,
                    const float* add3, int scale_mode, void* stream);

/* ZoomMask + ZoomImageWithFactor + Concat(/255) fused: X (B,H,W,8) NHWC network input
 * (deepIM_flownet.py:53-60).  The four z_* NCHW outputs are optional (all or none). */
int dim_zoom_net_input(const float* image_observed, const float* image_rendered, const float* mask_observed,
                       const float* mask_rendered, const float* zoom_factor, float* X_nhwc8, int B, int H, int W,
                       const float* means3, float* z_image_observed, float* z_image_rendered, float* z_mask_observed,
                       float* z_mask_rendered, void* stream);
/* the other input arities of get_convs (deepIM_flownet.py:33-66): mode 0 = masks (as dim_zoom_net_input), 1 = images only (INPUT_MASK
 * off, ZoomImage path: channels 6, 7 of X are zero), 2 = depth_observed / depth_rendered in place of the masks (INPUT_DEPTH without
 * masks: ZoomDepth's plain bilinear sample, / 255), 3 = the two zoomed masks alone in lanes 0, 1 (lanes 2-7 zero; images not read): the
 * second 8-lane group of the 10-channel first layer (INPUT_DEPTH with masks; the first group is mode 2).  X stays (B,H,W,8) NHWC. */
int dim_zoom_net_input_ex(const float* image_observed, const float* image_rendered, const float* extra_observed, const float* extra_rendered,
                          const float* zoom_factor, float* X_nhwc8, int B, int H, int W, const float* means3, int mode, void* stream);

/* ZoomTrans (zoom_trans.py:22-76): mode 0 copy, 1 (dx,dy)/wx, 2 (dx,dy)*wx */
int dim_zoom_trans(const float* zoom_factor, const float* in, float* out, int B, int mode, void* stream);

/* ---------------------------------------------------------------- SE(3)
 * rot_coord: 0 MODEL, 1 CAMERA, 2 CAMERA_NEW, 3 NAIVE.
 * pose_out = RT_transform(pose_src, se3[:, :4], se3[:, 4:])  (RT_transform.py:135-161); float64 inside. */
int dim_se3_compose(const float* pose_src, const float* se3, float* pose_out, double* pose_out_f64, int B, int rot_coord,
                    const float* T_means3, const float* T_stds3, void* stream);
/* (rot_quat, trans) = calc_RT_delta(pose_src, pose_tgt, rot_type="QUAT")  (RT_transform.py:16-48) */
int dim_se3_delta(const float* pose_src, const float* pose_tgt, float* rot_quat, float* trans, int B, int rot_coord,
                  const float* T_means3, const float* T_stds3, void* stream);
/* the same residual with the rotation as a 3x3 matrix (calc_RT_delta(..., rot_type="MATRIX"), RT_transform.py:16-48): rot_mat (B,3,3) */
int dim_se3_delta_matrix(const float* pose_src, const float* pose_tgt, float* rot_mat, float* trans, int B, int rot_coord,
                         const float* T_means3, const float* T_stds3, void* stream);
/* EULER deltas (RT_transform.py:139-140, :39-40: euler2mat / mat2euler with their default static-xyz axes): euler_trans6 (B,6) =
 * [ai, aj, ak, tx, ty, tz]; rot_euler (B,3). */
int dim_se3_compose_euler(const float* pose_src, const float* euler_trans6, float* pose_out, double* pose_out_f64, int B, int rot_coord,
                          const float* T_means3, const float* T_stds3, void* stream);
int dim_se3_delta_euler(const float* pose_src, const float* pose_tgt, float* rot_euler, float* trans, int B, int rot_coord,
                        const float* T_means3, const float* T_stds3, void* stream);
/* KT (B,3,4) = K * calc_se3(pose_src, pose_tgt): the per-sample matrix dim_depth_to_flow needs (batch_updater_py_multi.py:306-312) */
int dim_pose_to_KT(const float* pose_src, const float* pose_tgt, const float* K9, float* KT, int B, void* stream);
/* Transform3D custom op (transform3d.py:42-327); points/out/out_grad are (B,3,Npts). */
int dim_transform3d_fwd(const float* points, const float* rot, const float* trans, const float* pose_src, float* out, int B,
                        int Npts, int rot_coord, const float* T_means3, const float* T_stds3, void* stream);
int dim_transform3d_bwd(const float* out_grad, const float* points, const float* rot, const float* trans, const float* pose_src,
                        float* d_rot, float* d_trans, int B, int Npts, int rot_coord, const float* T_means3, const float* T_stds3,
                        void* stream);

/* ---------------------------------------------------------------- depth -> flow labels
 * device-pointer version of _flow (lib/flow_c/gpu_flow.hpp:1-3): flow (B,2,H,W) in (dy,dx), valid (B,1,H,W). */
int dim_depth_to_flow(const float* depth_src, const float* depth_tgt, const float* KT, const float* Kinv9, int B, int H, int W,
                      float* flow, float* valid, void* stream);
/* Test-time flow error of the first forward (deepim/core/tester.py:500-512; calc_EPE_one_pair :719-736 over the [flow, visible, bg]
 * list of par_generate_gt :706-716).  flow_pred = the network's flow_est_crop_output (B,2,H,W), rounded to float16 first as
 * tester.py:485-487 stores it; flow_gt (B,2,H,W) / visible (B,1,H,W) = calc_flow's outputs (dim_calc_flow_labels, weight_type 1);
 * bg = visible == 0 and depth_rendered == 0.  sums (B,5) float64 (device) = {epe_all, epe_viz, epe_vizbg, num_viz, num_vizbg} per
 * sample (num_all = H W), overwritten or -- accumulate != 0 -- added to.  Float64 arithmetic like numpy's (calc_flow returns
 * float64); deterministic two-stage sum.  workspace: dim_flow_epe_workspace_bytes(B). */
long dim_flow_epe_workspace_bytes(int B);
int dim_flow_epe_sums(const float* flow_pred, const float* flow_gt, const float* visible, const float* depth_rendered, int B, int H,
                      int W, void* workspace, double* sums, int accumulate, void* stream);

/* ---------------------------------------------------------------- data layer (test batches from raw file pixels)
 * The loader uploads what the image files hold -- obs_bgr / ren_bgr (B,H,W,3) uint8 in B,G,R order (cv2.IMREAD_COLOR), depth_rendered
 * (B,H,W) uint16 = metres * depth_factor -- and the blobs of get_data_pair_test_batch are built on the device:
 * image_observed / image_rendered (B,3,H,W) = RGB planes minus PIXEL_MEANS (given in config order B,G,R: lib/utils/image.py:709-720),
 * mask_rendered (B,1,H,W) = depth with values above mask_thr replaced by 1 (:478-488), bbox (B,4) {min_x,max_x,min_y,max_y} of
 * depth > mask_thr for dim_box_mask (TEST.INIT_MASK box_rendered, :437-460).  Any output (and its input) may be NULL.  W % 4 == 0. */
int dim_test_blobs_from_raw(const unsigned char* obs_bgr, const unsigned char* ren_bgr, const unsigned short* depth_rendered, int B, int H,
                            int W, float depth_factor, const float* pixel_means_bgr3, float mask_thr, float* image_observed,
                            float* image_rendered, float* mask_rendered, int* bbox, void* stream);

/* The general form for training AND test batches: reference lib/pair_matching/data_pair.py:22-72 (test) / :144-265 (train) through
 * lib/utils/image.py get_pair_image :65-183, get_gt_observed_depth :186-207, get_pair_depth :210-269, get_pair_mask :272-491.
 * Raw inputs, all (B,H,W[,3]) as the files hold them, any may be NULL (its outputs are then skipped):
 *   obs_bgr / ren_bgr uint8 BGR; bg_bgr uint8 BGR = the VOC background already fitted to (H,W) (image.py:125-165) -- pasted where the
 *   label image is 0 for samples with use_bg[b] != 0 (use_bg NULL: all), :166-173; depth_ren / depth_a / depth_b uint16 = metres *
 *   depth_factor; label uint8 label image with mask_idx (B) the object's label value.
 * Outputs: image_observed / image_rendered (B,3,H,W); mask_rendered (B,1,H,W) = depth with > mask_thr replaced by 1; depth_rendered,
 * depth_a_out, depth_b_out (B,1,H,W) metres; mask_label = (label == mask_idx), label_raw = (float)label (TRAIN.INIT_MASK 'mask_gt' feeds
 * the RAW label image, image.py:315-316); bbox_ren / bbox_label (B,4) {min_x,max_x,min_y,max_y} of depth > mask_thr / of mask_label for
 * dim_box_mask (INIT_MASK box_rendered / box_gt / box_gt_observed / box_).  W % 4 == 0. */
int dim_pair_blobs_from_raw(const unsigned char* obs_bgr, const unsigned char* bg_bgr, const int* use_bg, const unsigned char* ren_bgr,
                            const unsigned short* depth_ren, const unsigned short* depth_a, const unsigned short* depth_b,
                            const unsigned char* label, const int* mask_idx, int B, int H, int W, float depth_factor,
                            const float* pixel_means_bgr3, float mask_thr, float* image_observed, float* image_rendered,
                            float* mask_rendered, float* depth_rendered, float* depth_a_out, float* depth_b_out, float* mask_label,
                            float* label_raw, int* bbox_ren, int* bbox_label, void* stream);
/* mask_dilate (lib/utils/mask_dilate.py:10-55) with the random draws made by the caller: thickness4 (B,4) = {down, up, right, left}
 * displacement of the boundary copies, 0 = side skipped.  mask_out != mask_in. */
int dim_mask_dilate(const float* mask_in, const int* thickness4, float* mask_out, int B, int H, int W, void* stream);
/* First-iteration flow labels = calc_flow (lib/pair_matching/flow.py:12-81; float64 per pixel like numpy) + the weights of
 * get_pair_flow (image.py:531-545).  depth_src = rendered depth, depth_tgt = observed depth (B,1,H,W) metres; P12 (B,3,4) float64 (device) =
 * K se3_mul(pose_tgt, se3_inverse(pose_src)) as the reference forms it on the host; Kinv9_f64 = inv(K) in float64 (host pointer).
 * weight_type 0 all / 1 viz / 2 valid; flow (B,2,H,W) in "[h, w]" order unless standard_rep; flow_weights (B,2,H,W) or NULL. */
int dim_calc_flow_labels(const float* depth_src, const float* depth_tgt, const double* P12, const double* Kinv9_f64, int B, int H, int W,
                         double thresh, int standard_rep, int weight_type, float* flow, float* flow_weights, void* stream);
/* Point-matching labels (image.py:559-600): model[b,:,j] = table[table_off[b] + idx[b,j]] (idx < 0: zero-padded slot, weight 0),
 * weights (B,3,n), observed = R_obs model + t_obs.  table (N,3) = the points.xyz of all classes concatenated. */
int dim_point_clouds(const float* table, const int* table_off, const int* idx, const float* pose_observed, int B, int n, float* model,
                     float* weights, float* observed, void* stream);

/* ---------------------------------------------------------------- rasteriser
 * Mesh table in HBM: verts (sumV,3), uvs (sumV,2), faces (sumF,3 int32, indices local to the mesh),
 * mesh_table (C,4 int32) = {vert_off, nvert, face_off, nface}; textures = concatenated uint8 RGB images
 * (row 0 = top of texture_map.png), tex_table (C,3 int32) = {byte_off, Ht, Wt}.
 * class_index (B int32) in [0, n_classes), poses (B,3,4).  workspace: dim_raster_workspace_bytes(), 8-byte aligned (16 for the fast
 * resolve); its first 256 bytes are a header that remembers "the z-buffer behind me is clear" from one render to the next (no clear
 * pass): the OWNER ZEROES THOSE 256 BYTES when the memory is allocated or has been used for anything else.  (header, z-buffer, projected vertices,
 * covered-pixel list).  status (B int32, may be NULL): DIM_STATUS_BAD_CLASS / DIM_STATUS_BAD_FACE are OR-ed in.
 * Outputs (each may be NULL): image (B,3,H,W) = RGB - plane_means3 (the next iteration's image_rendered
 * blob), depth (B,1,H,W) metres, mask (B,1,H,W) = depth > mask_thr, bgr (B,H,W,3) as Render_Py.render
 * returns it, bbox (B,4) of the mask.   Replaces render_py_multi.py:112-147 + tester.py:563-578. */
long dim_raster_workspace_bytes(int B, int vmax, int H, int W);
int dim_raster_render(const float* verts, const float* uvs, const int* faces, const int* mesh_table, int n_classes, int vmax, int fmax,
                      const unsigned char* textures, const int* tex_table, const int* class_index, const float* poses,
                      const float* K9, int B, int H, int W, float znear, float zfar, int tex_bilinear, const float* plane_means3,
                      float mask_thr, void* workspace, float* image, float* depth, float* mask, float* bgr, int* bbox, int* status,
                      void* stream);
/* Lit variant of dim_raster_render: Render_Py_Light_ModelNet_Multi.render
 * (lib/render_glumpy/render_py_light_modelnet_multi.py:36-77 fragment shader, :153-235 render).
 * normals: per-vertex normals in the same table as verts.  light_pos (B,3) in GL camera coordinates, light_int (B,3)
 * (device pointers).  Colour = texel/255 * ((1-ratio) + ratio*clamp(cos(normal, light-position),0,1)) * light_int,
 * clamped to [0,1] and quantised to 8 bits (round to nearest) like the GL framebuffer. */
int dim_raster_render_lit(const float* verts, const float* normals, const float* uvs, const int* faces, const int* mesh_table,
                          int n_classes, int vmax, int fmax, const unsigned char* textures, const int* tex_table, const int* class_index,
                          const float* poses, const float* K9, int B, int H, int W, float znear, float zfar, int tex_bilinear,
                          const float* light_pos, const float* light_int, float brightness_ratio, const float* plane_means3,
                          float mask_thr, void* workspace, float* image, float* depth, float* mask, float* bgr, int* bbox, int* status,
                          void* stream);
/* The same render -- unlit when normals is NULL -- for a caller that re-renders into planes it knows: clean_bbox (B,4) {min x, max x,
 * min y, max y} (the bbox array a previous render into the SAME image / depth / mask / bgr planes returned; a different array than
 * `bbox`) promises that those planes hold background (image = -plane_means, depth = mask = bgr = 0) outside the box; pixels outside it
 * that this render does not cover are then not written again (the refinement loop's 2nd and 3rd render: tester.py:563-590 re-renders
 * the same object a few pixels away).  NULL = dim_raster_render / dim_raster_render_lit. */
int dim_raster_render_dirty(const float* verts, const float* normals, const float* uvs, const int* faces, const int* mesh_table,
                            int n_classes, int vmax, int fmax, const unsigned char* textures, const int* tex_table, const int* class_index,
                            const float* poses, const float* K9, int B, int H, int W, float znear, float zfar, int tex_bilinear,
                            const float* light_pos, const float* light_int, float brightness_ratio, const float* plane_means3,
                            float mask_thr, void* workspace, float* image, float* depth, float* mask, float* bgr, int* bbox, int* status,
                            const int* clean_bbox, void* stream);

/* deepim/core/tester.py:204-225 (and batch_updater_py_multi.py:233-255): light_pos[b] = 0.5*(dx,dy,dz) + (tx,-ty,-tz) of poses[b]. */
int dim_modelnet_light_position(const float* poses, float dx, float dy, float dz, float* light_pos, int B, void* stream);
/* mask[b] = rectangle [y0:y1, x0:x1] (end-exclusive) of bbox[b]  (data_pair.py:103-114, UPDATE_MASK box_rendered) */
/* bbox_of_mask (optional, (B,4)): bbox {min_x,max_x,min_y,max_y} of the rectangle just written, in dim_mask_bbox's convention --
 * the next iteration's ZoomMask then needs no scan of mask_observed. */
int dim_box_mask(const int* bbox, float* mask, int B, int H, int W, int* bbox_of_mask, void* stream);

/* ---------------------------------------------------------------- convolution stack (NHWC, f32 MFMA)
 * weights: pack once from the reference layout (Cout,Cin,KH,KW).  Cin must be 8 or a multiple of 32,
 * Cout a multiple of 64.  y = LeakyReLU_slope(conv(x) + bias); slope 1.0 = linear.
 * splits > 1 = split-K through `workspace` (dim_conv2d_workspace_floats), deterministic reduce.
 * tile: 0 auto | 1 128x128 (4 waves) | 2 128x64 | 3 64x64 | 4 128x128 (8 waves) (GEMM rows x output channels per workgroup of the
 *       gathered-tap kernel) | 6 the LDS-halo first-layer kernel (Cin 8, 7x7 / stride 2, Cout 64, dense f32 output).
 * Outputs are stored through a buffer descriptor with 32-bit byte offsets: N*OH*OW*out_cstride*4 must stay below 2^31. */
long dim_conv2d_packed_weight_floats(int Cout, int Cin, int KH, int KW);
int dim_conv2d_pack_weight(const float* w_oihw, float* w_packed, int Cout, int Cin, int KH, int KW, void* stream);
long dim_conv2d_workspace_floats(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int splits);
int dim_conv2d_fwd(const float* x, const float* w_packed, const float* bias, float* y, float* workspace, int N, int H, int W,
                   int Cin, int Cout, int KH, int KW, int stride, int pad, float slope, int splits, int tile, void* stream);
/* the two phases of a split-K dim_conv2d_fwd as separate calls (same kernels; lets a caller time / overlap them):
 * partial: raw K-slice sums -> workspace[splits][M][Cout];  reduce: y = LeakyReLU(sum_k slabs + bias). */
int dim_conv2d_fwd_partial(const float* x, const float* w_packed, float* workspace, int N, int H, int W, int Cin, int Cout, int KH,
                           int KW, int stride, int pad, int splits, int tile, void* stream);
int dim_splitk_reduce(const float* slabs, const float* bias, float* y, long M, int Cout, int splits, float slope, void* stream);
/* generalised dim_conv2d_fwd (no split-K): input pixel stride `in_cstride` (>= Cin), output written at channel offset
 * `out_coff` of rows `out_cstride` wide (concat buffers), and -- when osy > 0 -- scattered to
 * (oy,ox) = (ho*osy + ooy, wo*osx + oox) inside an OH x OW map (used for the phases of a deconvolution + Crop). */
/* Ho/Wo > 0: explicit output grid (asymmetric padding); pad_w >= 0: horizontal padding differs from pad; accumulate: y += result */
/* splits == 0 in dim_conv2d_fwd = "auto": all workgroups of a launch are equal, so a launch takes ceil(tiles / CUs) tile-times
 * on the busiest CU (1200 tiles on 256 CUs = 4.69 -> 5: 6 % of the chip idles).  Auto runs k*CUs tiles as one launch and the rest
 * as a split-K launch of proportionally shorter workgroups + reduce.  Needs workspace = dim_conv2d_workspace_floats(..., splits=0).
 * dim_conv2d_tail_plan reports the plan for a shape: first tile of the tail and its split count (1 = single launch). */
int dim_conv2d_tail_plan(int M, int Cout, int Cin, int KH, int KW, int tile, int* tail_begin_tile, int* tail_splits);
/* The default launch plans of the f32 forward path, in ONE place for every host (FlowNetHip in Python, dim_refiner_create in C):
 * (tile, splits) of a direct layer with M GEMM rows and `nchunks` K chunks of 32 (splits 0 = the "auto" tail plan above), and the
 * workgroup tile of a Winograd layer's plane GEMMs (3 / 4 / 5 = 64 x 64 / 128 x 128 / 128 x 256) for `tiles` transform tiles. */
int dim_conv_auto_plan(long M, int Cout, int nchunks, int cin, int* tile, int* splits);
int dim_winograd_gemm_tile(int Cout, long tiles);                          /* = ..._planes(Cout, tiles, 36) */
int dim_winograd_gemm_tile_planes(int Cout, long tiles, int planes);      /* planes: 36 (F(4x4,3x3), 5x5 / s2) or 81 (3x3 / s2) */
/* Arithmetic of the Winograd layers' plane GEMMs.  1 (default): every f32 operand enters the matrix pipe as the exact sum of three
 * bf16 terms and a product keeps the six largest term products, accumulated in f32 (error <= 3 * 2^-27 per product, below f32's own
 * rounding of the sums); 0: v_mfma_f32_32x32x2_f32 on the f32 operands.  Every packed Winograd weight buffer carries both images
 * (dim_winograd*_packed_weight_floats counts them).  Takes effect when a layer is next planned (launch or graph capture). */
int dim_set_winograd_split(int on);
int dim_get_winograd_split(void);
int dim_conv2d_fwd_ex(const float* x, const float* w_packed, const float* bias, float* y, int N, int H, int W, int Cin, int in_cstride,
                      int Cout, int KH, int KW, int stride, int pad, float slope, int tile, int out_cstride, int out_coff, int OH,
                      int OW, int osy, int osx, int ooy, int oox, int Ho, int Wo, int pad_w, int accumulate, void* stream);

/* Winograd form of the 3x3 / stride 1 / pad 1 layers (conv3_1, conv4_1, conv5_1, conv6_1 of
 * deepim/symbols/deepIM_flownet.py:103-191): same result as dim_conv2d_fwd up to f32 rounding (tests: 1e-4 relative).
 * m = output tile edge: 2 = F(2x2,3x3), 16 GEMMs, 2.25x fewer multiply-adds; 4 = F(4x4,3x3), 36 GEMMs, 4x fewer (Cook-Toom points
 * {0, 1, -1, 2, -1/2, inf}).  Weights are transformed once (dim_winograd_pack_weight from the (Cout,Cin,3,3) array, same m);
 * workspace = dim_winograd_workspace_floats floats.  in_cstride / out_cstride 0 = dense; y = LeakyReLU_slope(conv + bias). */
long dim_winograd_packed_weight_floats(int Cout, int Cin, int m);
long dim_winograd_workspace_floats(int N, int H, int W, int Cin, int Cout, int m);
int dim_winograd_pack_weight(const float* w_oihw, float* w_packed, int Cout, int Cin, int m, void* stream);
/* transformed weights of the INPUT gradient of the same layer (a Winograd convolution of dY with the flipped, transposed kernel: Cin
 * output and Cout input channels), read straight from the forward (Cout, Cin, 3, 3) array; same size as dim_winograd_pack_weight's */
int dim_winograd_dgrad_pack_weight(const float* w_oihw, float* w_packed, int Cout, int Cin, int m, void* stream);
/* events4: NULL, or four hipEvent_t recorded on the stream before the input transform, before / after the batched GEMM and after
 * the output transform (how bench.py times the GEMM launch separately from the transforms). */
int dim_conv2d_fwd_winograd(const float* x, const float* w_packed, const float* bias, float* y, float* workspace, int N, int H, int W,
                            int Cin, int in_cstride, int Cout, int out_cstride, int out_coff, float slope, int tile, int m,
                            void** events4, void* stream);
/* 5x5 / stride 2 / pad 2 layers (conv2, conv3 of deepim/symbols/deepIM_flownet.py:95-101) through Winograd as well: the layer is
 * the sum of four 3x3 / stride-1 / pad-1 convolutions of the phase images x[2r + py][2q + px] with the sub-kernels
 * w[2u + py][2v + px]; the four F(4x4,3x3)-transformed phase tiles are concatenated along the channels, so 36 GEMMs with K = 4 Cin
 * replace the 25-tap direct form (2.78x fewer multiply-adds) and the output transform is the one of the stride-1 layers.
 * Weights: MXNet layout (Cout,Cin,5,5).  Output (N, ceil(H/2), ceil(W/2), Cout).  Same f32 accuracy class as F(4x4,3x3). */
long dim_winograd5x5s2_packed_weight_floats(int Cout, int Cin);
long dim_winograd5x5s2_workspace_floats(int N, int H, int W, int Cin, int Cout);
int dim_winograd5x5s2_pack_weight(const float* w_oihw, float* w_packed, int Cout, int Cin, void* stream);
int dim_conv2d_fwd_winograd5x5s2(const float* x, const float* w_packed, const float* bias, float* y, float* workspace, int N, int H, int W,
                                 int Cin, int in_cstride, int Cout, int out_cstride, int out_coff, float slope, int tile, void** events4,
                                 void* stream);
/* Input gradient of the same 5x5 / stride-2 layers (training): dx (N,H,W,dx_cstride) = conv_transpose(dy (N,ceil(H/2),ceil(W/2),
 * dy_cstride), w) as ONE F(4x4,3x3) transform of dy, 36 GEMMs with K = Cout and N = 4 Cin (the four phase images of dx), and a
 * phase-scattering output transform; dx is overwritten.  Weights: dim_winograd5x5s2_dgrad_pack_weight from the same (Cout,Cin,5,5)
 * array (dim_winograd5x5s2_packed_weight_floats floats); workspace: dim_winograd5x5s2_workspace_floats floats. */
int dim_winograd5x5s2_dgrad_pack_weight(const float* w_oihw, float* w_packed, int Cout, int Cin, void* stream);
int dim_conv2d_dgrad_winograd5x5s2(const float* dy, const float* w_packed, float* dx, float* workspace, int N, int H, int W, int Cin,
                                   int dx_cstride, int Cout, int dy_cstride, int tile, void* stream);
/* Weight gradient of the same layers through Winograd (training): dw (Cout,Cin,k,k) MXNet layout (=, or += when accumulate) =
 * scale * sum over pixels of x (*) dy.  S = 1: 3x3 / stride 1 / pad 1 (k = 3); S = 2: 5x5 / stride 2 / pad 2 (k = 5, four phase
 * images of x).  Per 4x4 tile of dy this is the correlation F(3x3,4x4): V = B^T x B (the forward's input transform), D = G4 dy G4^T,
 * 36 plane products contracted over tiles and batch on the wgrad MFMA kernel (`splits` pixel ranges, deterministic slab reduce),
 * dW = A3^T dM A3: 4x (S = 1) / 2.78x (S = 2) fewer multiply-adds than dim_conv2d_wgrad.  x (N,H,W,in_cstride), dy (N,Ho,Wo,dy_cstride). */
long dim_conv2d_wgrad_winograd_workspace_floats(int N, int H, int W, int Cin, int Cout, int S, int splits);
int dim_conv2d_wgrad_winograd(const float* x, const float* dy, float* dw_oihw, float* workspace, int N, int H, int W, int Cin, int in_cstride,
                              int Cout, int dy_cstride, int S, int splits, float scale, int accumulate, void* stream);
/* 3x3 / stride-2 / pad-1 layers (conv4, conv5: deepim/symbols/deepIM_flownet.py:118-126, 143-151) in the Winograd domain through their
 * phase images: per axis the even phase meets one tap (F(4,1) = identity), the odd phase two (F(4,2), points {0, 1, -1, 2, inf}): 81
 * multiplies per 4 x 4 output tile and channel pair against 144, 81 plane GEMMs on the stream-K kernel of the other Winograd layers.
 * x (N,H,W,in_cstride)[:Cin] -> y (N,Ho,Wo,out_cstride)[out_coff:+Cout], Ho = floor((H - 1) / 2) + 1; Cin % 32 == 0, Cout % 64 == 0.
 * Same result as dim_conv2d_fwd(3, 3, stride 2, pad 1) up to f32 rounding (1.4e-6 of max |y| at Cin = 256; direct: 3e-7). */
long dim_winograd3x3s2_packed_weight_floats(int Cout, int Cin);
int dim_winograd3x3s2_use(int H, int W, int Cin, int Cout);   /* 1: the launch plan sends this layer (input map H x W) through this path */
long dim_winograd3x3s2_workspace_floats(int N, int H, int W, int Cin, int Cout);
int dim_winograd3x3s2_pack_weight(const float* w_oihw, float* w_packed, int Cout, int Cin, void* stream);
int dim_conv2d_fwd_winograd3x3s2(const float* x, const float* w_packed, const float* bias, float* y, float* workspace, int N, int H, int W,
                                 int Cin, int in_cstride, int Cout, int out_cstride, int out_coff, float slope, int tile, void** events4,
                                 void* stream);
/* fc6 (deepim/symbols/deepIM_flownet.py:196-198) for the small batches of the refinement loop as a weight stream: y (B,Out) =
 * LeakyReLU_slope(x (B,H,W,C NHWC) . W^T + bias) with W packed by dim_fc_pack_weight.  Every workgroup owns a contiguous range of K
 * chunks and all outputs and writes one partial tile into `workspace` (dim_fc_fwd_workspace_floats floats); a second kernel sums the
 * partials in a fixed order.  32 batch rows per pass.  Same result as dim_conv2d_fwd(KH=H, KW=W) up to the summation order. */
long dim_fc_fwd_workspace_floats(int C, int H, int W, int Out);
int dim_fc_fwd(const float* x, const float* w_packed, const float* bias, float* y, float* workspace, int B, int C, int H, int W, int Out,
               float slope, void* stream);
/* dim_conv2d_pack_weight with the output channels zero-padded to CoutPad (multiple of 64) */
int dim_conv2d_pack_weight_padded(const float* w_oihw, float* w_packed, int Cout, int CoutPad, int Cin, int KH, int KW, void* stream);
/* Decoder (deepIM_flownet.py:213-299): y[..., out_coff:out_coff+Cout] = LeakyReLU(Crop(Deconvolution(x, k=4, s=2, p=0) + bias,
 * offset=(crop,crop))) as four 2x2 sub-pixel convolutions on the MFMA kernel.  Weight: MXNet layout (Cin, Cout, 4, 4). */
long dim_deconv4x4s2_packed_weight_floats(int Cin, int Cout);
int dim_deconv4x4s2_pack_weight(const float* w_iohw, float* w_packed, int Cin, int Cout, void* stream);
int dim_deconv4x4s2_fwd(const float* x, const float* w_packed, const float* bias, float* y, int N, int H, int W, int Cin,
                        int in_cstride, int Cout, int OH, int OW, int crop, float slope, int out_cstride, int out_coff, int tile,
                        void* stream);
/* same operator for a tiny channel count (upsample_flow6to5 / 5to4, 2 -> 2), direct; weight in MXNet layout, unpacked */
int dim_deconv4x4s2_tiny_fwd(const float* x, const float* w_iohw, const float* bias, float* y, int N, int H, int W, int Cin,
                             int in_cstride, int Cout, int OH, int OW, int crop, int out_cstride, int out_coff, void* stream);
/* 3x3 (any KHxKW, stride 1) convolution to 1 or 2 channels (Convolution1/2/3, mask_conv3), no activation.
 * Weight (Cout,Cin,KH,KW) packed to [Cout][KH][KW][CinPad], CinPad = Cin rounded up to 32 (floats: Cout*KH*KW*CinPad). */
int dim_conv_small_cout_pack_weight(const float* w_oihw, float* w_packed, int Cout, int Cin, int KH, int KW, void* stream);
int dim_conv_small_cout_fwd(const float* x, const float* w_packed, const float* bias, float* y, int N, int H, int W, int Cin,
                            int in_cstride, int Cout, int KH, int KW, int pad, int out_cstride, int out_coff, void* stream);
/* Deconvolution(k=32, s=16, num_group=C, no bias) + Crop(offset (crop,crop)) -> NCHW planes (N,C,OH,OW), times `scale`;
 * mode 1 applies a sigmoid (mask probability).  deepIM_flownet.py:326-340, :513-529, :845-872. */
int dim_upsample16_fwd(const float* x_nhwc, const float* w_c1_32_32, float* y_nchw, int N, int C, int h, int w, int OH, int OW,
                       int crop, float scale, int mode, void* stream);
/* ---------------------------------------------------------------- backward of the convolution stack (training)
 * dgrad: dx[..., :Cin] (+)= d/dx of y = conv(x, W (Cout,Cin,KH,KW), stride 1|2, pad); runs on the forward MFMA kernel with the
 * weights re-packed by dim_conv2d_dgrad_pack_weight (stride 2 = four input phases).  dx's channel stride must be >= Cin
 * rounded up to 64. */
long dim_conv2d_dgrad_packed_weight_floats(int Cout, int Cin, int KH, int KW, int stride, int pad);
int dim_conv2d_dgrad_pack_weight(const float* w_oihw, float* w_packed, int Cout, int Cin, int KH, int KW, int stride, int pad,
                                 void* stream);
int dim_conv2d_dgrad(const float* dy, const float* w_dgrad_packed, float* dx, int N, int H, int W, int Cin, int dx_cstride, int Ho,
                     int Wo, int Cout, int dy_cstride, int KH, int KW, int stride, int pad, int accumulate, int tile, void* stream);
/* wgrad: dw_packed (same layout as dim_conv2d_pack_weight's output, so SGD updates the packed weights in place)
 * (+)= sum over pixels of dz (N,Ho,Wo,dz_cstride)[dz_coff:+Cout] x gathered x (N,H,W,in_cstride)[:Cin]. */
long dim_conv2d_wgrad_workspace_floats(int Cout, int Cin, int KH, int KW, int splits);
int dim_conv2d_wgrad(const float* x, const float* dz, float* dw_packed, float* workspace, int N, int H, int W, int Cin, int in_cstride,
                     int Ho, int Wo, int Cout, int dz_cstride, int dz_coff, int KH, int KW, int stride, int pad, int splits,
                     int accumulate, void* stream);
/* The same weight gradient delivered in the MXNet layout dw_oihw (Cout_rows, Cin, KH, KW) = (accumulate ? dw_oihw : 0) + scale * dW (rows
 * Cout_rows .. Cout - 1 of a channel-padded gradient are dropped): the pixel-split slabs stay in `workspace` --
 * dim_conv2d_wgrad_workspace_floats(Cout, Cin, KH, KW, splits + 1) floats, also for splits == 1 -- and the layout converter sums
 * them in slab order on its way out, so the packed intermediate, its round trip through HBM and the reduce launch of
 * dim_conv2d_wgrad + dim_conv2d_unpack_weight disappear; same bits as that pair whenever it sums slabs serially (fewer than 64 slabs or
 * slabs of >= 2^18 floats; otherwise the lane-parallel reduce runs first, as there).  bf16_mfma: products on the bf16 matrix pipe
 * (dim_conv2d_wgrad_bf16).  Replaces the weight-gradient half of the executor's backward (deepim/core/module.py:1205-1213). */
int dim_conv2d_wgrad_oihw(const float* x, const float* dz, float* dw_oihw, float* workspace, int N, int H, int W, int Cin, int in_cstride,
                          int Ho, int Wo, int Cout, int dz_cstride, int dz_coff, int KH, int KW, int stride, int pad, int splits,
                          int bf16_mfma, int Cout_rows, float scale, int accumulate, void* stream);
/* ---------------------------------------------------------------- bf16 matrix pipe (training mode, BASELINE configs[2])
 * The reference trains in fp32 (deepim/train.py:338-414); these twins of the convolution entry points run the same implicit GEMMs
 * on v_mfma_f32_32x32x16_bf16 (f32 accumulate, 16x the f32 matrix rate): activations and gradients stay fp32 in HBM and are rounded
 * to bf16 (nearest even) on the way into LDS; `w_packed_bf16` is the bf16 image -- dim_f32_to_bf16, element by element -- of the
 * SAME packed array the f32 entry point takes (dim_conv2d_pack_weight, dim_conv2d_dgrad_pack_weight, dim_deconv4x4s2_pack_weight,
 * dim_conv2d_pack_weight_padded), so every packer is shared.  Arguments otherwise as the f32 functions (dim_conv2d_fwd_bf16:
 * splits >= 1, no "auto" mode).  dim_conv2d_wgrad_bf16 returns an fp32 gradient in the packed layout, like dim_conv2d_wgrad.
 * Declared tolerance vs the fp32 path: 2^-8 relative per product, i.e. ~4e-3 L2-relative on a layer output / gradient tensor.
 * Tile 6 (the 8-channel 7x7 / stride-2 / 64-filter first layer from an LDS-resident patch) exists on both pipes.
 * bf16-only tiles: 7 = LDS-halo kernel (8 x 16 pixels x 128 channels; 3x3 / 5x5, stride 1 / 2, dense output); 8 = 128 x 256 gathered taps;
 * 9 = stride-1 patch kernel (16 x 16 pixels x 128 or 64 channels, KH, KW <= 3 with 2 .. 9 taps, Cin % 32 == 0, Cout % 64 == 0, dense or
 * scattered output, batched phases); dim_conv2d_dgrad_bf16 with tile 9 applies it to every phase of a strided gradient that has >= 2
 * taps and runs the single-tap phase on the gathered-tap kernel. */
/* The input gradient of a layer whose INPUT is another layer's LeakyReLU output, with that LeakyReLU' and the lower layer's bias
 * gradient folded into the epilogue (tile 9, the bf16 patch kernel: 3x3 / stride 1 and 5x5 / stride 2 layers on maps of >= 1200 pixels):
 * dz (N,H,W,Cin) = dX * (y_act > 0 ? 1 : slope), db[Cin] (+)= column sums of dz.  y_act (N,H,W,Cin) is the stored activation; Cin % 64 == 0.
 * Same dz bits as dim_conv2d_dgrad_bf16(tile 9) followed by dim_lrelu_bwd_bias_grad; db differs in summation order only.  Replaces one
 * 12-bytes-per-element pass over the gradient per layer (deepim/core/module.py:1205-1213 runs it inside MXNet's backward). */
long dim_conv2d_dgrad_lrelu_workspace_floats(int N, int H, int W, int Cin, int stride);
int dim_conv2d_dgrad_bf16_lrelu(const float* dy, const void* w_dgrad_packed_bf16, float* dz, const float* y_act, float slope, float* db,
                                float* workspace, int N, int H, int W, int Cin, int dx_cstride, int Ho, int Wo, int Cout, int dy_cstride,
                                int KH, int KW, int stride, int pad, int accumulate_db, void* stream);
int dim_f32_to_bf16(const float* src, void* dst_bf16, long n, void* stream);
int dim_bf16_to_f32(const void* src_bf16, float* dst, long n, void* stream);
/* The packers with the rounding folded in: each writes exactly dim_f32_to_bf16 of its f32 twin's output (same element count, same
 * order) in one pass over the MXNet-layout weights -- the training executor re-packs every layer after every update.
 * dim_conv2d_pack_weight_bf16 covers both dim_conv2d_pack_weight (CoutPad == Cout) and dim_conv2d_pack_weight_padded. */
int dim_conv2d_pack_weight_bf16(const float* w_oihw, void* w_packed_bf16, int Cout, int CoutPad, int Cin, int KH, int KW, void* stream);
int dim_conv2d_dgrad_pack_weight_bf16(const float* w_oihw, void* w_packed_bf16, int Cout, int Cin, int KH, int KW, int stride, int pad,
                                      void* stream);
int dim_deconv4x4s2_pack_weight_bf16(const float* w_iohw, void* w_packed_bf16, int Cin, int Cout, void* stream);
int dim_fc_dgrad_pack_weight_bf16(const float* w_out_in, void* w_packed_bf16, int Out, int C, int H, int W, void* stream);
int dim_conv2d_fwd_bf16(const float* x, const void* w_packed_bf16, const float* bias, float* y, float* workspace, int N, int H, int W,
                        int Cin, int Cout, int KH, int KW, int stride, int pad, float slope, int splits, int tile, void* stream);
int dim_conv2d_fwd_ex_bf16(const float* x, const void* w_packed_bf16, const float* bias, float* y, int N, int H, int W, int Cin,
                           int in_cstride, int Cout, int KH, int KW, int stride, int pad, float slope, int tile, int out_cstride,
                           int out_coff, int OH, int OW, int osy, int osx, int ooy, int oox, int Ho, int Wo, int pad_w, int accumulate,
                           void* stream);
int dim_conv2d_dgrad_bf16(const float* dy, const void* w_dgrad_packed_bf16, float* dx, int N, int H, int W, int Cin, int dx_cstride,
                          int Ho, int Wo, int Cout, int dy_cstride, int KH, int KW, int stride, int pad, int accumulate, int tile,
                          void* stream);
/* The same with the contraction (taps x output channels of dy) cut into `splits` ranges (tile 3 / 4 only): the small maps give too
 * few tiles to fill the chip.  Partial sums go to `workspace` (dim_conv2d_dgrad_splitk_workspace_floats: `splits` copies of dx) and
 * one pass adds them up in a fixed order (deterministic); every phase of a strided gradient needs >= splits K chunks. */
long dim_conv2d_dgrad_splitk_workspace_floats(int N, int H, int W, int dx_cstride, int splits);
int dim_conv2d_dgrad_bf16_splitk(const float* dy, const void* w_dgrad_packed_bf16, float* dx, float* workspace, int N, int H, int W, int Cin,
                                 int dx_cstride, int Ho, int Wo, int Cout, int dy_cstride, int KH, int KW, int stride, int pad,
                                 int accumulate, int tile, int splits, void* stream);
int dim_deconv4x4s2_fwd_bf16(const float* x, const void* w_packed_bf16, const float* bias, float* y, int N, int H, int W, int Cin,
                             int in_cstride, int Cout, int OH, int OW, int crop, float slope, int out_cstride, int out_coff, int tile,
                             void* stream);
/* pixel-split count that makes dim_conv2d_wgrad_bf16's whole grid resident at once on n_cu compute units (size the slab workspace
 * with it: dim_conv2d_wgrad_workspace_floats) */
int dim_conv2d_wgrad_bf16_splits(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int n_cu);
int dim_conv2d_wgrad_bf16(const float* x, const float* dz, float* dw_packed, float* workspace, int N, int H, int W, int Cin, int in_cstride,
                          int Ho, int Wo, int Cout, int dz_cstride, int dz_coff, int KH, int KW, int stride, int pad, int splits,
                          int accumulate, void* stream);
/* db[c] (+)= sum_m dz[m][dz_coff + c]   (workspace: dim_bias_grad_workspace_floats) */
long dim_bias_grad_workspace_floats(int M, int C);
int dim_bias_grad(const float* dz, float* db, float* workspace, int M, int C, int dz_cstride, int dz_coff, int accumulate, void* stream);
/* LeakyReLU backward in place on a channel range: dy *= (y > 0 ? 1 : slope) */
int dim_lrelu_bwd(const float* y, int y_cstride, int y_coff, float* dy, int dy_cstride, int dy_coff, long M, int C, float slope,
                  void* stream);
/* both of the above in one pass over the gradient map: dy *= (y > 0 ? 1 : slope) in place, db[c] (+)= sum_m dy[m][dy_coff + c]
 * (the pair MXNet's LeakyReLU backward + Convolution backward-bias make; deepIM_flownet.py:67-208) */
long dim_lrelu_bwd_bias_grad_workspace_floats(int M, int C);
int dim_lrelu_bwd_bias_grad(const float* y, int y_cstride, int y_coff, float* dy, int dy_cstride, int dy_coff, float* db, float* workspace,
                            int M, int C, float slope, int accumulate, void* stream);
/* ---------------------------------------------------------------- training-only pieces (csrc/train.hip)
 * layout converters: packed conv / fc weights (or gradients) back to the MXNet layouts; fc6 dgrad weights */
int dim_conv2d_unpack_weight(const float* w_packed, float* w_oihw, int Cout, int CoutPad, int Cin, int KH, int KW, float scale,
                             int accumulate, void* stream);
int dim_fc_unpack_weight(const float* w_packed, float* w_out_in, int Out, int C, int H, int W, void* stream);
int dim_fc_dgrad_pack_weight(const float* w_out_in, float* w_packed, int Out, int C, int H, int W, void* stream);
/* loss gradients (get_loss, deepIM_flownet.py:303-560); loss_sum (1 float, may be NULL) accumulates the un-scaled loss for metrics */
int dim_flow_loss_grad(const float* flow_est, const float* flow_label, const float* flow_weights, float* grad, long n, float normalize_flow,
                       float grad_scale, float* loss_sum, void* stream);
int dim_logistic_grad(const float* logits, const float* label, float* grad, float* prob, long n, float grad_scale_over_num_output,
                      void* stream);
int dim_pm_l1_grad(const float* p_est, const float* p_obs, const float* weights, float* grad, long n, float norm_term, float grad_scale,
                   float* loss_sum, void* stream);
/* the point-matching loss with SE3_PM_LOSS_TYPE 'L1' (0, = dim_pm_l1_grad) | 'L2' (1) | 'smooth_L1' (2, mx.sym.smooth_l1 with
 * scalar = SE3_PM_SL1_SCALAR)   (deepIM_flownet.py:458-499) */
int dim_pm_loss_grad(const float* p_est, const float* p_obs, const float* weights, float* grad, long n, float norm_term, float grad_scale,
                     int loss_type, float smooth_l1_scalar, float* loss_sum, void* stream);
/* SE3_DIST_LOSS (deepIM_flownet.py:396-437): rot_loss = 1 - (rot_gt . rot_est_norm)^2 with grad_scale LW_ROT and trans_loss =
 * TRANS_LOSS_TYPE(zoom_trans_est - zoom_trans_gt) with grad_scale LW_TRANS; zoom_trans_est = trans_w fc7 + trans_b is recomputed from
 * fc7 (B,256).  The gradients are ADDED to d_rot_norm (B,4) / d_zoom_trans (B,3), the inputs of dim_pose_head_bwd; loss_sums2 (2
 * floats, may be NULL) accumulate the un-scaled rot / trans loss sums (metrics Rot_L2Loss / Trans_L2Loss). */
int dim_se3_dist_loss_grad(const float* rot_est_norm, const float* rot_gt, const float* fc7, const float* trans_w, const float* trans_b,
                           const float* zoom_trans_gt, float* d_rot_norm, float* d_zoom_trans, int B, float lw_rot, float lw_trans,
                           int trans_loss_type, float smooth_l1_scalar, float* loss_sums2, void* stream);
/* L2Normalization(instance, eps 1e-10) of the quaternion head; backward of the pose head down to dz6 (B,256) */
int dim_quat_normalize(const float* rot, float* rot_norm, int B, void* stream);
int dim_pose_head_bwd(const float* fc6a, const float* fc7, const float* rot_raw, const float* d_rot_norm, const float* d_trans,
                      const float* fc7_w, const float* rot_w, const float* trans_w, float* d_rot, float* dz7, float* dz6, int B,
                      void* stream);
int dim_fc_wgrad(const float* dz, const float* x, float* dW, float* db, int B, int Out, int In, void* stream);
/* fc6's weight gradient straight in the MXNet layout: dW (Out, C*H*W flattened (c, h, w)) = dz (B, Out)^T . x (B, H, W, C NHWC), for batches
 * of 1 .. 32 rows, C % 16 == 0, H * W <= 80 (fc6: 8 x 10 x 1024 -> 256).  f32 products, summed over the batch in row order.  Replaces
 * dim_conv2d_wgrad(KH = H, KW = W) + dim_fc_unpack_weight: one 84 MB write instead of a packed intermediate and its conversion. */
int dim_fc_wgrad_nhwc(const float* dz, const float* x, float* dW, int B, int Out, int C, int H, int W, void* stream);
int dim_upsample16_bwd(const float* dout_nchw, const float* w_c1_32_32, float* df_nhwc, int N, int C, int h, int w, int OH, int OW, int crop,
                       float scale, void* stream);
long dim_conv_small_cout_bwd_workspace_floats(int N, int H, int W, int Cin, int Cout, int KH, int KW);
int dim_conv_small_cout_bwd(const float* x, const float* dy, const float* w_oihw, float* dx, float* dw_oihw, float* db, float* workspace,
                            int N, int H, int W, int Cin, int in_cstride, int dx_cstride, int Cout, int KH, int KW, int pad,
                            int accumulate_dx, void* stream);
int dim_deconv4x4s2_tiny_bwd(const float* x, int x_cstride, const float* dy, int dy_cstride, int dy_coff, const float* w_iohw, float* dx,
                             float* dw_iohw, float* db, int N, int H, int W, int Cin, int Cout, int OH, int OW, int crop, void* stream);
/* mx.optimizer.SGD: mom = momentum*mom - lr*(rescale_grad*g + wd*w); w += mom */
int dim_sgd_momentum(float* w, const float* grad, float* mom, long n, float lr, float momentum, float wd, float rescale_grad, void* stream);
/* mx.optimizer.Adam step (deepim/train.py:338-375 selects it with TRAIN.optimizer == "adam"): lr_t = lr*sqrt(1-beta2^t)/(1-beta1^t)
 * is computed by the caller from the update count t; mean / var are the two state tensors. */
int dim_adam(float* w, const float* grad, float* mean, float* var, long n, float lr_t, float beta1, float beta2, float epsilon, float wd,
             float rescale_grad, void* stream);
/* FullyConnected weight (Out, C*H*W) [mx Flatten order c,h,w] -> [(h,w,c)][Out] so fc6 is dim_conv2d_fwd
 * with KH=H, KW=W on the NHWC feature map. */
int dim_fc_pack_weight(const float* w_out_in, float* w_packed, int Out, int C, int H, int W, void* stream);
/* fc7 + LeakyReLU(0.1) + rot(4) + trans(3) + inverse ZoomTrans -> se3 (B,7)  (deepIM_flownet.py:203-208,:956-971).
 * weights in the reference (out,in) layout; fc7_out (B,256) optional. */
int dim_pose_head_fwd(const float* fc6, const float* fc7_w, const float* fc7_b, const float* rot_w, const float* rot_b,
                      const float* trans_w, const float* trans_b, const float* zoom_factor, float* se3, float* fc7_out, int B,
                      void* stream);

/* ---------------------------------------------------------------- resident refinement loop (hosts without torch)
 * The inner loop of pred_eval (deepim/core/tester.py:476-598) for a batch of B 480x640 pairs that stays in HBM, FAST_TEST graph,
 * UPDATE_MASK 'box_rendered': per iteration ZoomMask + ZoomImageWithFactor + Concat -> encoder -> fc6/fc7/rot/trans -> RT_transform
 * -> render -> box mask.  dim_refiner_create packs the weights (MXNet-layout device arrays given by name: <layer>_weight / _bias of
 * the ten encoder layers, fc6, fc7, rot, trans; the bias / fc7 / rot / trans arrays are read in place and must outlive the object),
 * allocates every buffer and fixes the launch plan; dim_refiner_run only enqueues kernels on `stream` (no allocation, no sync: it
 * may be captured into a hipGraph) and never writes its six input blobs.  Outputs: poses_iter (test_iter,B,3,4), se3_iter
 * (test_iter,B,7), status_iter (test_iter,B) with the DIM_STATUS_* bits.  The mesh table pointers must outlive the object. */
typedef struct dim_refiner dim_refiner;
typedef struct {
  int B, H, W, test_iter;
  float K9[9];
  float pixel_means_bgr[3]; /* config.network.PIXEL_MEANS, in its B,G,R order */
  float T_means[3], T_stds[3];
  int rot_coord; /* 0 MODEL, 1 CAMERA, 2 CAMERA_NEW, 3 NAIVE */
  float znear, zfar;
  int tex_bilinear;
  const float* verts;
  const float* uvs;
  const int* faces;
  const int* mesh_table;
  int n_classes, vmax, fmax;
  const unsigned char* textures;
  const int* tex_table;
} dim_refiner_desc;
int dim_refiner_create(dim_refiner** out, const dim_refiner_desc* desc, const char* const* param_names, const float* const* param_ptrs,
                       int n_params, void* stream);
int dim_refiner_run(dim_refiner* r, const float* image_observed, const float* image_rendered, const float* mask_observed,
                    const float* mask_rendered, const float* src_pose, const int* class_index, float* poses_iter, float* se3_iter,
                    int* status_iter, void* stream);
int dim_refiner_destroy(dim_refiner* r);

#ifdef __cplusplus
}
#endif
#endif /* DEEPIM_HIP_H_ */
