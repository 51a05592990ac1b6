// On-device depth/RGB rasteriser: replaces the glumpy/OpenGL renderer and its glReadPixels
// round trip (/root/reference/lib/render_glumpy/render_py_multi.py:89-147, projection :152-169,
// view matrix :171-178) so the refinement loop's render -> mask -> next-iteration-input step
// (deepim/core/tester.py:563-590, lib/pair_matching/data_pair.py:86-135) never leaves HBM.
//
// Conventions pinned by the reference: pixel (i,j) samples the pinhole projection at
// (u,v) = (fx*X/Z+cx, fy*Y/Z+cy) = (i,j)  [u0 = K[0,2]+0.5 with GL pixel centres at +0.5];
// depth output = camera-frame Z in metres, 0 for background, clipped to [znear, zfar];
// colour = unlit texture lookup, texture_map.png stored with row 0 = top (v is flipped).
// Implementation-defined in GL and fixed here (same as oracle/raster.c): 8-bit sub-pixel
// snapping, top-left fill rule, nearest or bilinear clamp-to-edge texel filter, z-fight ties
// broken by the lower face index.
// Near plane (GL clips primitives against zNear = 0.25, render_py_multi.py:152-169): a triangle with all vertices at Z >= zNear
// takes the screen-space path (its fragments outside [znear, zfar] are discarded per pixel = what clipping gives for it); one wholly
// in front of the plane is dropped; one that straddles it is clipped in CAMERA space (Sutherland-Hodgman, cut points computed from
// the inside to the outside vertex so neighbours agree), rasterised as a fan, and shaded with camera-space barycentrics of the
// original triangle -- its screen triangle does not exist when a vertex is behind the eye.  A refinement that diverges toward the
// camera therefore renders what GL would, instead of losing every triangle that crosses Z = 0.
//
// Passes (all on one stream): clear z-buffer (u64 = depth bits << 32 | face id) -> project
// vertices -> one thread per triangle, atomicMin per covered pixel -> resolve (textured RGB
// written straight into the next iteration's network-input blobs + mask + bbox).
#include "common.h"

namespace dim {

struct Edges {
  long long A[3], B[3], C[3];
  long long area;
  bool tl[3];
};

__device__ inline int snap_px(float u) { return (int)floorf(u * 256.0f + 0.5f); }

__device__ inline bool setup_edges(const int X[3], const int Y[3], Edges& e) {
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    int a = (i + 1) % 3, b = (i + 2) % 3;
    e.A[i] = (long long)Y[a] - (long long)Y[b];
    e.B[i] = (long long)X[b] - (long long)X[a];
    e.C[i] = (long long)X[a] * (long long)Y[b] - (long long)X[b] * (long long)Y[a];
  }
  e.area = e.A[0] * X[0] + e.B[0] * Y[0] + e.C[0];
  if (e.area == 0) return false;
  if (e.area < 0) {
#pragma unroll
    for (int i = 0; i < 3; ++i) { e.A[i] = -e.A[i]; e.B[i] = -e.B[i]; e.C[i] = -e.C[i]; }
    e.area = -e.area;
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) e.tl[i] = (e.A[i] > 0) || (e.A[i] == 0 && e.B[i] > 0);
  return true;
}

__device__ inline bool inside_tri(const Edges& e, long long px, long long py, long long E[3]) {
  bool in = true;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    E[i] = e.A[i] * px + e.B[i] * py + e.C[i];
    in = in && (E[i] > 0 || (E[i] == 0 && e.tl[i]));
  }
  return in;
}

struct MeshRef {
  int vert_off, nvert, face_off, nface;
};

// First 256 bytes of the workspace.  `magic` + the geometry say: "the z-buffer behind this header holds only clear keys" -- every render
// leaves it that way (the pass that consumes a covered key resets it), so the next render of the same geometry needs no clear pass
// (39 MB of stores per 16 frames, and a launch).  Anything else in these words (a fresh or recycled allocation whose first 256 bytes
// the owner zeroed, another batch size on the same memory) makes the vertex pass clear the whole z-buffer first.
struct ResolveHdr {
  unsigned count;  // covered pixels appended so far
  unsigned magic0, magic1;
  int B, H, W, vmax;
  unsigned pad[57];
};
constexpr unsigned kZbMagic0 = 0x44494d5au, kZbMagic1 = 0x62756621u;
__device__ __forceinline__ bool zbuf_known_clear(const ResolveHdr* hdr, int B, int H, int W, int vmax) {
  return hdr->magic0 == kZbMagic0 && hdr->magic1 == kZbMagic1 && hdr->B == B && hdr->H == H && hdr->W == W && hdr->vmax == vmax;
}

// scr[b][i] = (u, v, Zc) for vertex i of the sample's mesh
__global__ __launch_bounds__(256) void raster_vertex_kernel(const float* __restrict__ verts, const int* __restrict__ mesh_table,
                                                            const int* __restrict__ class_index, const float* __restrict__ poses,
                                                            float fx, float fy, float cx, float cy, int vmax, int n_classes,
                                                            int* __restrict__ status, float* __restrict__ scr, ResolveHdr* __restrict__ hdr,
                                                            unsigned long long* __restrict__ zbuf, int B, int H, int W, int* __restrict__ bbox,
                                                            int wide) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  // ---- what the separate clear / init launches did.  (The header is only READ here -- every block must see the same answer; the
  // triangle pass, which runs when all of these blocks are done, stamps it.)
  if (!zbuf_known_clear(hdr, B, H, W, vmax)) {
    const long n = (long)B * H * W, stride = (long)gridDim.x * gridDim.y * blockDim.x;
    const long k0 = ((long)blockIdx.y * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x;
    if (wide) {   // 16-byte stores (the workspace is 16-byte aligned, B H W even)
      ulonglong2* z2 = reinterpret_cast<ulonglong2*>(zbuf);
      for (long k = k0; k < n / 2; k += stride) z2[k] = make_ulonglong2(0xFFFFFFFFFFFFFFFFull, 0xFFFFFFFFFFFFFFFFull);
    } else {
      for (long k = k0; k < n; k += stride) zbuf[k] = 0xFFFFFFFFFFFFFFFFull;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (b == 0) hdr->count = 0;                                                          // empty covered-pixel list
    if (bbox) { bbox[4 * b + 0] = W; bbox[4 * b + 1] = -1; bbox[4 * b + 2] = H; bbox[4 * b + 3] = -1; }   // empty box (one-pass resolve)
  }
  const int cls = class_index[b];
  if ((unsigned)cls >= (unsigned)n_classes) {  // no such mesh: the sample renders as background and says so
    if (i == 0 && status) atomicOr(status + b, DIM_STATUS_BAD_CLASS);
    return;
  }
  const int* mt = mesh_table + 4 * cls;
  if (i >= mt[1]) return;
  const float* p = verts + 3 * (long)(mt[0] + i);
  const float* P = poses + 12 * b;
  float xc = __fmaf_rn(P[0], p[0], __fmaf_rn(P[1], p[1], __fmaf_rn(P[2], p[2], P[3])));
  float yc = __fmaf_rn(P[4], p[0], __fmaf_rn(P[5], p[1], __fmaf_rn(P[6], p[2], P[7])));
  float zc = __fmaf_rn(P[8], p[0], __fmaf_rn(P[9], p[1], __fmaf_rn(P[10], p[2], P[11])));
  float* s = scr + ((long)b * vmax + i) * 3;
  s[0] = __fmaf_rn(fx, __fdiv_rn(xc, zc), cx);
  s[1] = __fmaf_rn(fy, __fdiv_rn(yc, zc), cy);
  s[2] = zc;
}

__device__ inline bool load_tri(const float* scr_b, const int* face, int X[3], int Y[3], float iz[3]) {
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float* s = scr_b + 3 * (long)face[k];
    if (!(s[2] > 1e-6f) || !(fabsf(s[0]) < 1.0e6f) || !(fabsf(s[1]) < 1.0e6f)) return false;
    X[k] = snap_px(s[0]);
    Y[k] = snap_px(s[1]);
    iz[k] = __fdiv_rn(1.0f, s[2]);
  }
  return true;
}

__device__ inline float interp_z(const long long E[3], float inv_area, const float iz[3], float b[3]) {
  b[0] = (float)E[0] * inv_area;
  b[1] = (float)E[1] * inv_area;
  b[2] = (float)E[2] * inv_area;
  float invz = __fmaf_rn(b[2], iz[2], __fmaf_rn(b[1], iz[1], __fmul_rn(b[0], iz[0])));
  return __fdiv_rn(1.0f, invz);
}

constexpr float kZClipMin = 1.0e-4f;
constexpr float kCoordLim = 1.0e6f;

// camera-space position of vertex p under pose P (3x4 row-major): the vertex pass's arithmetic, bit for bit
__device__ inline void cam_point(const float* __restrict__ P, const float* __restrict__ p, float c[3]) {
  c[0] = __fmaf_rn(P[0], p[0], __fmaf_rn(P[1], p[1], __fmaf_rn(P[2], p[2], P[3])));
  c[1] = __fmaf_rn(P[4], p[0], __fmaf_rn(P[5], p[1], __fmaf_rn(P[6], p[2], P[7])));
  c[2] = __fmaf_rn(P[8], p[0], __fmaf_rn(P[9], p[1], __fmaf_rn(P[10], p[2], P[11])));
}

// Sutherland-Hodgman against z >= zc: 0, 3 or 4 vertices in out
__device__ inline int clip_near(const float cam[3][3], float zc, float out[4][3]) {
  int n = 0;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float* a = cam[i];
    const float* b = cam[(i + 1) % 3];
    const bool ina = a[2] >= zc, inb = b[2] >= zc;
    if (ina) { out[n][0] = a[0]; out[n][1] = a[1]; out[n][2] = a[2]; ++n; }
    if (ina != inb) {
      const float* pi = ina ? a : b;
      const float* po = ina ? b : a;
      const float tt = __fdiv_rn(__fsub_rn(zc, pi[2]), __fsub_rn(po[2], pi[2]));
      out[n][0] = __fmaf_rn(tt, __fsub_rn(po[0], pi[0]), pi[0]);
      out[n][1] = __fmaf_rn(tt, __fsub_rn(po[1], pi[1]), pi[1]);
      out[n][2] = zc;
      ++n;
    }
  }
  return n;
}

// affine (camera-space) barycentrics of the point pixel (x, y) sees at depth z, w.r.t. the triangle cam; float64, un-fused
// (-ffp-contract=off) like oracle/raster.c cam_bary
__device__ inline void cam_bary(const float cam[3][3], int x, int y, float z, float fx, float fy, float cx, float cy, float w[3]) {
  const double P[3] = {(double)z * (((double)x - (double)cx) / (double)fx), (double)z * (((double)y - (double)cy) / (double)fy), (double)z};
  double e1[3], e2[3], d0[3], d1[3], d2[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    e1[k] = (double)cam[1][k] - (double)cam[0][k];
    e2[k] = (double)cam[2][k] - (double)cam[0][k];
    d0[k] = (double)cam[0][k] - P[k];
    d1[k] = (double)cam[1][k] - P[k];
    d2[k] = (double)cam[2][k] - P[k];
  }
  const double n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
  const double nn = n[0] * n[0] + n[1] * n[1] + n[2] * n[2];
  const double c0[3] = {d1[1] * d2[2] - d1[2] * d2[1], d1[2] * d2[0] - d1[0] * d2[2], d1[0] * d2[1] - d1[1] * d2[0]};
  const double c1[3] = {d2[1] * d0[2] - d2[2] * d0[1], d2[2] * d0[0] - d2[0] * d0[2], d2[0] * d0[1] - d2[1] * d0[0]};
  const double b0 = (c0[0] * n[0] + c0[1] * n[1] + c0[2] * n[2]) / nn;
  const double b1 = (c1[0] * n[0] + c1[1] * n[1] + c1[2] * n[2]) / nn;
  w[0] = (float)b0;
  w[1] = (float)b1;
  w[2] = (float)(1.0 - b0 - b1);
}

// one screen triangle into the z-buffer; clamp_near: a fragment of a clipped piece may round below the plane it was cut at
__device__ inline void raster_one(const int X[3], const int Y[3], const float iz[3], int f, int H, int W, float znear, float zfar,
                                  bool clamp_near, float zc, unsigned long long* __restrict__ zb) {
  Edges e;
  if (!setup_edges(X, Y, e)) return;
  int minX = min(X[0], min(X[1], X[2])), maxX = max(X[0], max(X[1], X[2]));
  int minY = min(Y[0], min(Y[1], Y[2])), maxY = max(Y[0], max(Y[1], Y[2]));
  int x0 = max((minX + 255) >> 8, 0), x1 = min(maxX >> 8, W - 1);
  int y0 = max((minY + 255) >> 8, 0), y1 = min(maxY >> 8, H - 1);
  if (x0 > x1 || y0 > y1) return;
  const float inv_area = __fdiv_rn(1.0f, (float)e.area);
  for (int y = y0; y <= y1; ++y)
    for (int x = x0; x <= x1; ++x) {
      long long E[3];
      if (!inside_tri(e, (long long)x * 256, (long long)y * 256, E)) continue;
      float bw[3];
      float z = interp_z(E, inv_area, iz, bw);
      if (clamp_near && z < zc) z = zc;
      if (!(z >= znear && z <= zfar)) continue;
      unsigned long long key = ((unsigned long long)__float_as_uint(z) << 32) | (unsigned)f;
      atomicMin(zb + (long)y * W + x, key);
    }
}

__global__ __launch_bounds__(256) void raster_tri_kernel(const int* __restrict__ faces, const int* __restrict__ mesh_table,
                                                         const int* __restrict__ class_index, const float* __restrict__ scr,
                                                         const float* __restrict__ verts, const float* __restrict__ poses, float fx,
                                                         float fy, float cx, float cy, int vmax, int H, int W, float znear, float zfar,
                                                         int n_classes, unsigned long long* __restrict__ zbuf, ResolveHdr* __restrict__ hdr,
                                                         int B) {
  const int b = blockIdx.y;
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (b == 0 && f == 0) {   // the z-buffer was clear when this pass started; the resolve passes of this render restore that
    hdr->magic0 = kZbMagic0; hdr->magic1 = kZbMagic1; hdr->B = B; hdr->H = H; hdr->W = W; hdr->vmax = vmax;
  }
  const int cls = class_index[b];
  if ((unsigned)cls >= (unsigned)n_classes) return;
  const int* mt = mesh_table + 4 * cls;
  if (f >= mt[3]) return;
  const int* face = faces + 3 * (long)(mt[2] + f);
  const float* scr_b = scr + (long)b * vmax * 3;
  int X[3], Y[3];
  float iz[3];
  unsigned long long* zb = zbuf + (long)b * H * W;
  const float zc = fmaxf(znear, kZClipMin);
  const float vz0 = scr_b[3 * (long)face[0] + 2], vz1 = scr_b[3 * (long)face[1] + 2], vz2 = scr_b[3 * (long)face[2] + 2];
  const int nin = (int)(vz0 >= zc) + (int)(vz1 >= zc) + (int)(vz2 >= zc);
  if (nin == 0) return;  // wholly in front of the near plane (or NaN): clipped away
  if (nin < 3) {         // straddles the near plane: clip in camera space, rasterise the pieces (rare: a few triangles of a close object)
    float cam[3][3], poly[4][3], su[4], sv[4];
#pragma unroll
    for (int k = 0; k < 3; ++k) cam_point(poses + 12 * b, verts + 3 * (long)(mt[0] + face[k]), cam[k]);
    const int np = clip_near(cam, zc, poly);
    bool ok = np >= 3;
    for (int k = 0; k < np; ++k) {
      su[k] = __fmaf_rn(fx, __fdiv_rn(poly[k][0], poly[k][2]), cx);
      sv[k] = __fmaf_rn(fy, __fdiv_rn(poly[k][1], poly[k][2]), cy);
      ok = ok && (fabsf(su[k]) < kCoordLim) && (fabsf(sv[k]) < kCoordLim);
    }
    if (!ok) return;
    for (int piece = 0; piece + 2 < np; ++piece) {
      const int idx[3] = {0, piece + 1, piece + 2};
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        X[k] = snap_px(su[idx[k]]);
        Y[k] = snap_px(sv[idx[k]]);
        iz[k] = __fdiv_rn(1.0f, poly[idx[k]][2]);
      }
      raster_one(X, Y, iz, f, H, W, znear, zfar, true, zc, zb);
    }
    return;
  }
  if (!load_tri(scr_b, face, X, Y, iz)) return;
  raster_one(X, Y, iz, f, H, W, znear, zfar, false, zc, zb);
}

__device__ inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// One thread per pixel.  Outputs (any may be null):
//   image  (B,3,H,W): plane c = RGB[c] - plane_means[c]          (lib/utils/image.py:709-720 transform())
//   depth  (B,1,H,W) metres;  mask (B,1,H,W) = depth > mask_thr   (deepim/core/tester.py:575-577)
//   bgr    (B,H,W,3) 0..255 as Render_Py.render returns it
//   bbox   (B,4) {min_x,max_x,min_y,max_y} of mask (pre-initialised to {W,-1,H,-1})
// Lit (ModelNet) shading inputs, render_py_light_modelnet_multi.py:36-77: all null / unused when LIT is false
struct LitArgs {
  const float* verts;      // model-frame positions (same table as the vertex pass); also read for near-clipped faces
  const float* normals;    // per-vertex normals, same indexing as verts
  const float* poses;      // (B,3,4)
  const float* light_pos;  // (B,3) GL camera coordinates
  const float* light_int;  // (B,3)
  float ratio;             // brightness_ratio
  float fx, fy, cx, cy;    // pinhole (near-clipped faces: pixel -> camera-space point)
  float zclip;             // max(znear, kZClipMin): a face with a vertex in front of it was clipped by the triangle pass
};

// Colour of pixel (x, y) of sample b whose z-buffer key names face f (the heavy path: exact edge functions again, perspective-correct
// barycentrics, three dependent gathers, texel fetch, optional Lambert term).  Returns false for a face id outside the mesh.
template <bool LIT>
__device__ __forceinline__ bool shade_pixel(const LitArgs& lit, const float* __restrict__ uvs, const int* __restrict__ faces,
                                            const int* __restrict__ mesh_table, const unsigned char* __restrict__ tex,
                                            const int* __restrict__ tex_table, int cls, const float* __restrict__ scr_b, int b, int x, int y,
                                            unsigned f_id, float z_key, int tex_bilinear, float& r, float& g, float& bl) {
  const int* mt = mesh_table + 4 * cls;
  if (f_id >= (unsigned)mt[3]) return false;
  const int f = (int)f_id;
  const int* face = faces + 3 * (long)(mt[2] + f);
  int X[3], Y[3];
  float iz[3], tu[3], tv[3];
  bool clipped = false;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float* uv = uvs + 2 * (long)(mt[0] + face[k]);
    tu[k] = uv[0];
    tv[k] = uv[1];
    clipped = clipped || !(scr_b[3 * (long)face[k] + 2] >= lit.zclip);
  }
  float w0, w1, w2, z;  // attribute = (w2 a2 + (w1 a1 + w0 a0)) * z
  if (clipped) {        // a piece of a near-clipped triangle: camera-space barycentrics of the ORIGINAL triangle
    float cam[3][3], wb[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) cam_point(lit.poses + 12 * b, lit.verts + 3 * (long)(mt[0] + face[k]), cam[k]);
    cam_bary(cam, x, y, z_key, lit.fx, lit.fy, lit.cx, lit.cy, wb);
    w0 = wb[0]; w1 = wb[1]; w2 = wb[2]; z = 1.0f;
  } else {
    load_tri(scr_b, face, X, Y, iz);
    Edges e;
    long long E[3];
    setup_edges(X, Y, e);
    inside_tri(e, (long long)x * 256, (long long)y * 256, E);
    const float inv_area = __fdiv_rn(1.0f, (float)e.area);
    float bw[3];
    interp_z(E, inv_area, iz, bw);
    w0 = __fmul_rn(bw[0], iz[0]); w1 = __fmul_rn(bw[1], iz[1]); w2 = __fmul_rn(bw[2], iz[2]); z = z_key;
  }
  float u = __fmul_rn(__fmaf_rn(w2, tu[2], __fmaf_rn(w1, tu[1], __fmul_rn(w0, tu[0]))), z);
  float v = __fmul_rn(__fmaf_rn(w2, tv[2], __fmaf_rn(w1, tv[1], __fmul_rn(w0, tv[0]))), z);
  const int* tt = tex_table + 3 * cls;
  const unsigned char* T = tex + tt[0];
  const int Ht = tt[1], Wt = tt[2];
  if (!tex_bilinear) {
    int tx = clampi((int)floorf(__fmul_rn(u, (float)Wt)), 0, Wt - 1);
    int ty = clampi((int)floorf(__fmul_rn(v, (float)Ht)), 0, Ht - 1);
    const unsigned char* px = T + ((long)(Ht - 1 - ty) * Wt + tx) * 3;
    r = px[0]; g = px[1]; bl = px[2];
  } else {
    float xf = __fsub_rn(__fmul_rn(u, (float)Wt), 0.5f), yf = __fsub_rn(__fmul_rn(v, (float)Ht), 0.5f);
    float x0f = floorf(xf), y0f = floorf(yf);
    float ax = __fsub_rn(xf, x0f), ay = __fsub_rn(yf, y0f);
    int xa = clampi((int)x0f, 0, Wt - 1), xb = clampi((int)x0f + 1, 0, Wt - 1);
    int ya = clampi((int)y0f, 0, Ht - 1), yb = clampi((int)y0f + 1, 0, Ht - 1);
    const unsigned char* p00 = T + ((long)(Ht - 1 - ya) * Wt + xa) * 3;
    const unsigned char* p01 = T + ((long)(Ht - 1 - ya) * Wt + xb) * 3;
    const unsigned char* p10 = T + ((long)(Ht - 1 - yb) * Wt + xa) * 3;
    const unsigned char* p11 = T + ((long)(Ht - 1 - yb) * Wt + xb) * 3;
    float c[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      float top = __fmaf_rn(ax, __fsub_rn((float)p01[k], (float)p00[k]), (float)p00[k]);
      float bot = __fmaf_rn(ax, __fsub_rn((float)p11[k], (float)p10[k]), (float)p10[k]);
      c[k] = __fmaf_rn(ay, __fsub_rn(bot, top), top);
      if (!LIT) c[k] = floorf(c[k]);  // tester.py:244 astype("uint8")
    }
    r = c[0]; g = c[1]; bl = c[2];
  }
  if (LIT) {
    // perspective-correct varyings v_normal / v_position, then GL camera frame (y, z flipped)
    float n[3], p[3];
    const float* N0 = lit.normals + 3 * (long)(mt[0] + face[0]);
    const float* N1 = lit.normals + 3 * (long)(mt[0] + face[1]);
    const float* N2 = lit.normals + 3 * (long)(mt[0] + face[2]);
    const float* P0 = lit.verts + 3 * (long)(mt[0] + face[0]);
    const float* P1 = lit.verts + 3 * (long)(mt[0] + face[1]);
    const float* P2 = lit.verts + 3 * (long)(mt[0] + face[2]);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      n[k] = __fmul_rn(__fmaf_rn(w2, N2[k], __fmaf_rn(w1, N1[k], __fmul_rn(w0, N0[k]))), z);
      p[k] = __fmul_rn(__fmaf_rn(w2, P2[k], __fmaf_rn(w1, P1[k], __fmul_rn(w0, P0[k]))), z);
    }
    const float* P = lit.poses + 12 * b;
    float Ng[3], Pg[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float sgn = k == 0 ? 1.f : -1.f;
      Ng[k] = __fmul_rn(sgn, __fmaf_rn(P[4 * k + 2], n[2], __fmaf_rn(P[4 * k + 1], n[1], __fmul_rn(P[4 * k], n[0]))));
      Pg[k] = __fmul_rn(sgn, __fadd_rn(__fmaf_rn(P[4 * k + 2], p[2], __fmaf_rn(P[4 * k + 1], p[1], __fmul_rn(P[4 * k], p[0]))),
                                       P[4 * k + 3]));
    }
    const float* L = lit.light_pos + 3 * b;
    float sx = __fsub_rn(L[0], Pg[0]), sy = __fsub_rn(L[1], Pg[1]), sz = __fsub_rn(L[2], Pg[2]);
    float dotv = __fmaf_rn(Ng[2], sz, __fmaf_rn(Ng[1], sy, __fmul_rn(Ng[0], sx)));
    float ls = __fsqrt_rn(__fmaf_rn(sz, sz, __fmaf_rn(sy, sy, __fmul_rn(sx, sx))));
    float ln = __fsqrt_rn(__fmaf_rn(Ng[2], Ng[2], __fmaf_rn(Ng[1], Ng[1], __fmul_rn(Ng[0], Ng[0]))));
    float br = __fdiv_rn(dotv, __fmul_rn(ls, ln));
    br = fmaxf(fminf(br, 1.0f), 0.0f);
    float kk = __fmaf_rn(lit.ratio, br, __fsub_rn(1.0f, lit.ratio));
    const float* I = lit.light_int + 3 * b;
    float c0 = __fmul_rn(__fdiv_rn(r, 255.0f), __fmul_rn(kk, I[0]));
    float c1 = __fmul_rn(__fdiv_rn(g, 255.0f), __fmul_rn(kk, I[1]));
    float c2 = __fmul_rn(__fdiv_rn(bl, 255.0f), __fmul_rn(kk, I[2]));
    // 8-bit framebuffer: clamp to [0,1], round to nearest
    r = floorf(__fmaf_rn(fminf(fmaxf(c0, 0.f), 1.f), 255.0f, 0.5f));
    g = floorf(__fmaf_rn(fminf(fmaxf(c1, 0.f), 1.f), 255.0f, 0.5f));
    bl = floorf(__fmaf_rn(fminf(fmaxf(c2, 0.f), 1.f), 255.0f, 0.5f));
  }
  return true;
}

// ---- resolve, pass 1 (streaming): one thread per 4 consecutive pixels.  Everything that does not need the triangle is final after
// this pass -- depth and mask (the key carries z), the bbox, and the background colour of every pixel -- and the covered pixels are
// appended to a compact list (one atomicAdd per wave).  The first version resolved in one pass with one thread per pixel: an object
// covers 2-6 % of a 480x640 frame, so the edge set-up and the three dependent gathers ran on nearly empty waves (102 us per 16 frames
// with the object in view against 23 us without).
__global__ __launch_bounds__(256) void raster_resolve_stream_kernel(const unsigned long long* __restrict__ zbuf, int H, int W, float pm0,
                                                                    float pm1, float pm2, float mask_thr, float* __restrict__ image,
                                                                    float* __restrict__ depth, float* __restrict__ mask,
                                                                    float* __restrict__ bgr, int4* __restrict__ wave_ext,
                                                                    int waves_per_sample, ResolveHdr* __restrict__ hdr,
                                                                    unsigned* __restrict__ list, const int* __restrict__ clean) {
  const int b = blockIdx.y;
  const int plane = H * W;
  const int q = blockIdx.x * blockDim.x + threadIdx.x;  // quad index inside the image
  const int pix = 4 * q;
  const bool live = pix < plane;
  float z[4] = {0.f, 0.f, 0.f, 0.f};
  unsigned cov = 0;
  // `clean` (B,4) {min x, max x, min y, max y}: the caller's promise that the output planes already hold background outside that box
  // (the bbox the previous render into the same planes returned): a quad outside it with nothing covered is not written again
  bool in_dirty = true;
  if (clean && live) {
    const int y = pix / W, x0 = pix - y * W;
    const int* c = clean + 4 * b;
    in_dirty = y >= c[2] && y <= c[3] && x0 + 3 >= c[0] && x0 <= c[1];
  }
  if (live) {
    const ulonglong2* zp = reinterpret_cast<const ulonglong2*>(zbuf + (long)b * plane + pix);
    const ulonglong2 k01 = zp[0], k23 = zp[1];
    const unsigned long long key[4] = {k01.x, k01.y, k23.x, k23.y};
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (key[k] != 0xFFFFFFFFFFFFFFFFull) {
        z[k] = __uint_as_float((unsigned)(key[k] >> 32));
        cov |= 1u << k;
      }
    const long o = (long)b * plane + pix;
    const bool wr = in_dirty || cov != 0;
    if (depth && wr) *reinterpret_cast<float4*>(depth + o) = make_float4(z[0], z[1], z[2], z[3]);
    if (mask && wr)
      *reinterpret_cast<float4*>(mask + o) = make_float4(z[0] > mask_thr ? 1.f : 0.f, z[1] > mask_thr ? 1.f : 0.f, z[2] > mask_thr ? 1.f : 0.f,
                                                         z[3] > mask_thr ? 1.f : 0.f);
    if (image && wr) {  // background everywhere; pass 2 overwrites the covered pixels
      float* im = image + (long)b * 3 * plane + pix;
      *reinterpret_cast<float4*>(im) = make_float4(-pm0, -pm0, -pm0, -pm0);
      *reinterpret_cast<float4*>(im + plane) = make_float4(-pm1, -pm1, -pm1, -pm1);
      *reinterpret_cast<float4*>(im + 2 * (long)plane) = make_float4(-pm2, -pm2, -pm2, -pm2);
    }
    if (bgr && wr) {
      float4* qd = reinterpret_cast<float4*>(bgr + o * 3);
      qd[0] = qd[1] = qd[2] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  const unsigned long long any = __ballot(cov != 0);
  const int lane = threadIdx.x & 63;
  // this wave's slot in the per-sample extent table (bbox of the mask; see raster_bbox_reduce_kernel)
  int4* ext = wave_ext ? wave_ext + (long)b * waves_per_sample + ((blockIdx.x * blockDim.x + threadIdx.x) >> 6) : nullptr;
  if (any == 0) {  // wave-uniform: nothing covered
    if (ext && lane == 0) *ext = make_int4(0x7FFFFFFF, -1, 0x7FFFFFFF, -1);
    return;
  }
  // ---- append: exclusive prefix of popcount(cov) over the wave from three ballots (the count has three bits)
  const unsigned n = __popc(cov);
  unsigned before = 0, total = 0;
#pragma unroll
  for (int bit = 0; bit < 3; ++bit) {
    const unsigned long long m = __ballot((n >> bit) & 1u);
    before += __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u)) << bit;
    total += (unsigned)__popcll(m) << bit;
  }
  unsigned base = 0;
  if (lane == 0) base = atomicAdd(&hdr->count, total);
  base = __builtin_amdgcn_readfirstlane(base);
  unsigned slot = base + before;
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (cov & (1u << k)) list[slot++] = (unsigned)(b * plane + pix + k);
  // ---- bbox of the mask (z > mask_thr): per-lane extent of its quad, min / max over the wave, ONE PLAIN STORE per wave into its slot
  // of the extent table; raster_bbox_reduce_kernel folds the table.  (The first version did atomicMin / atomicMax on bbox[4 b + i]
  // from every covered wave: ~200 atomics per address and launch, which L2 retires one after the other at ~175 ns each -- 35 of the
  // 62 us of this pass, found by switching the block off.)
  if (ext) {
    const int y = pix / W, x0 = pix - y * W;
    int lo = 0x7FFFFFFF, hi = -1, ylo = 0x7FFFFFFF, yhi = -1;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (z[k] > mask_thr) {
        lo = min(lo, x0 + k);
        hi = max(hi, x0 + k);
        ylo = yhi = y;
      }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      lo = min(lo, __shfl_xor(lo, d));
      hi = max(hi, __shfl_xor(hi, d));
      ylo = min(ylo, __shfl_xor(ylo, d));
      yhi = max(yhi, __shfl_xor(yhi, d));
    }
    if (lane == 0) *ext = make_int4(lo, hi, ylo, yhi);
  }
}

// bbox[b] = (min x, max x, min y, max y) over the wave extents of sample b; (W, -1, H, -1) when the mask is empty, as raster_init_kernel
// leaves it for the one-pass path
__device__ __forceinline__ void raster_bbox_reduce(const int4* __restrict__ wave_ext, int waves_per_sample, int H, int W,
                                                   int* __restrict__ bbox, int b) {
  int lo = 0x7FFFFFFF, hi = -1, ylo = 0x7FFFFFFF, yhi = -1;
  for (int i = threadIdx.x; i < waves_per_sample; i += 256) {
    const int4 e = wave_ext[(long)b * waves_per_sample + i];
    lo = min(lo, e.x); hi = max(hi, e.y); ylo = min(ylo, e.z); yhi = max(yhi, e.w);
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    lo = min(lo, __shfl_xor(lo, d));
    hi = max(hi, __shfl_xor(hi, d));
    ylo = min(ylo, __shfl_xor(ylo, d));
    yhi = max(yhi, __shfl_xor(yhi, d));
  }
  __shared__ int red[4][4];
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = lo; red[threadIdx.x >> 6][1] = hi; red[threadIdx.x >> 6][2] = ylo; red[threadIdx.x >> 6][3] = yhi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w) { lo = min(lo, red[w][0]); hi = max(hi, red[w][1]); ylo = min(ylo, red[w][2]); yhi = max(yhi, red[w][3]); }
    bbox[4 * b + 0] = min(lo, W); bbox[4 * b + 1] = hi; bbox[4 * b + 2] = min(ylo, H); bbox[4 * b + 3] = yhi;
  }
}

// ---- resolve, pass 2: one thread per LISTED pixel (full waves), grid-stride over the list whose length lives on the device
// The last pass of a render.  SHADE = false: no colour output was asked for -- only the two duties below.
//   * the first B blocks fold the wave extents of the stream pass into bbox[b] (was a launch of its own);
//   * EVERY listed key is reset to "clear" by the thread that consumed it, so the z-buffer is clear again when the render ends and the
//     next one needs no clear pass (ResolveHdr).
template <bool LIT, bool SHADE>
__global__ __launch_bounds__(256) void raster_resolve_shade_kernel(LitArgs lit, const float* __restrict__ uvs, const int* __restrict__ faces,
                                                                   const int* __restrict__ mesh_table,
                                                                   const unsigned char* __restrict__ tex, const int* __restrict__ tex_table,
                                                                   const int* __restrict__ class_index, const float* __restrict__ scr,
                                                                   unsigned long long* __restrict__ zbuf, int vmax, int H, int W,
                                                                   int tex_bilinear, float pm0, float pm1, float pm2,
                                                                   float* __restrict__ image, float* __restrict__ bgr,
                                                                   int* __restrict__ status, const ResolveHdr* __restrict__ hdr,
                                                                   const unsigned* __restrict__ list, const int4* __restrict__ wave_ext,
                                                                   int waves_per_sample, int B, int* __restrict__ bbox) {
  if (bbox && (int)blockIdx.x < B) raster_bbox_reduce(wave_ext, waves_per_sample, H, W, bbox, (int)blockIdx.x);
  const unsigned n = hdr->count;
  const unsigned plane = (unsigned)(H * W);
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const unsigned e = list[i];
    const unsigned long long key = zbuf[e];
    zbuf[e] = 0xFFFFFFFFFFFFFFFFull;
    if (!SHADE) continue;
    const int b = (int)(e / plane);
    const unsigned pix = e - (unsigned)b * plane;
    const int y = (int)(pix / (unsigned)W), x = (int)(pix - (unsigned)y * (unsigned)W);
    const float z = __uint_as_float((unsigned)(key >> 32));
    float r = 0.f, g = 0.f, bl = 0.f;
    // (class_index was range-checked by the vertex pass: a sample with a bad class has no covered pixel)
    const bool ok = shade_pixel<LIT>(lit, uvs, faces, mesh_table, tex, tex_table, class_index[b], scr + (long)b * vmax * 3, b, x, y,
                                     (unsigned)(key & 0xFFFFFFFFu), z, tex_bilinear, r, g, bl);
    if (!ok && status) atomicOr(status + b, DIM_STATUS_BAD_FACE);  // a corrupt key: the pixel keeps z but is coloured black, loudly
    if (image) {
      float* im = image + (long)b * 3 * plane + pix;
      im[0] = r - pm0;
      im[plane] = g - pm1;
      im[2 * (long)plane] = bl - pm2;
    }
    if (bgr) {
      float* qd = bgr + ((long)b * plane + pix) * 3;
      qd[0] = bl; qd[1] = g; qd[2] = r;
    }
  }
}

// One-pass resolve (one thread per pixel): kept for image widths that are not a multiple of 4.
template <bool LIT>
__global__ __launch_bounds__(256) void raster_resolve_kernel(LitArgs lit, const float* __restrict__ uvs, const int* __restrict__ faces,
                                                             const int* __restrict__ mesh_table, const unsigned char* __restrict__ tex,
                                                             const int* __restrict__ tex_table, const int* __restrict__ class_index,
                                                             const float* __restrict__ scr, unsigned long long* __restrict__ zbuf,
                                                             int vmax, int H, int W, int tex_bilinear, float pm0, float pm1, float pm2,
                                                             float mask_thr, float* __restrict__ image, float* __restrict__ depth,
                                                             float* __restrict__ mask, float* __restrict__ bgr, int* __restrict__ bbox,
                                                             int* __restrict__ status) {
  const int b = blockIdx.z;
  const int y = blockIdx.y;
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const long plane = (long)H * W;
  float r = 0.f, g = 0.f, bl = 0.f, z = 0.f;
  bool in_img = x < W;
  if (in_img) {
    unsigned long long key = zbuf[(long)b * plane + (long)y * W + x];
    if (key != 0xFFFFFFFFFFFFFFFFull) {
      zbuf[(long)b * plane + (long)y * W + x] = 0xFFFFFFFFFFFFFFFFull;   // leave the z-buffer clear (ResolveHdr)
      z = __uint_as_float((unsigned)(key >> 32));
      const bool ok = shade_pixel<LIT>(lit, uvs, faces, mesh_table, tex, tex_table, class_index[b], scr + (long)b * vmax * 3, b, x, y,
                                       (unsigned)(key & 0xFFFFFFFFu), z, tex_bilinear, r, g, bl);
      if (!ok && status) atomicOr(status + b, DIM_STATUS_BAD_FACE);
    }
    const long o = (long)y * W + x;
    if (image) {
      float* im = image + (long)b * 3 * plane;
      im[o] = r - pm0;
      im[plane + o] = g - pm1;
      im[2 * plane + o] = bl - pm2;
    }
    if (depth) depth[(long)b * plane + o] = z;
    if (mask) mask[(long)b * plane + o] = z > mask_thr ? 1.f : 0.f;
    if (bgr) {
      float* q = bgr + ((long)b * plane + o) * 3;
      q[0] = bl; q[1] = g; q[2] = r;
    }
  }
  if (bbox) {
    bool on = in_img && z > mask_thr;
    unsigned long long ball = __ballot(on);
    // per-wave min/max x via ballot (lanes are consecutive x), one atomic set per wave with coverage
    if (ball) {
      const int lane = threadIdx.x & 63;
      if (lane == 0) {
        int wave_x0 = blockIdx.x * blockDim.x + (threadIdx.x & ~63);
        int lo = __ffsll((long long)ball) - 1;
        int hi = 63 - __clzll((long long)ball);
        atomicMin(&bbox[4 * b + 0], wave_x0 + lo);
        atomicMax(&bbox[4 * b + 1], wave_x0 + hi);
        atomicMin(&bbox[4 * b + 2], y);
        atomicMax(&bbox[4 * b + 3], y);
      }
    }
  }
}

// mask[b] = filled rectangle [y0:y1, x0:x1] (END-EXCLUSIVE: lib/pair_matching/data_pair.py:103-114)
// of bbox[b]; empty bbox -> all zeros (the reference would raise in np.min; status reports it).
__global__ __launch_bounds__(256) void box_mask_kernel(const int* __restrict__ bbox, float* __restrict__ mask, int H, int W,
                                                       int* __restrict__ bbox_of_mask) {
  const int b = blockIdx.z, y = blockIdx.y;
  const int x4 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (x4 >= W) return;
  const int* bb = bbox + 4 * b;
  const int xs = bb[0], xe = bb[1], ys = bb[2], ye = bb[3];
  if (bbox_of_mask && y == 0 && x4 == 0) {
    // bbox {min_x,max_x,min_y,max_y} of the rectangle written below, so that the next ZoomMask need not scan the mask for it:
    // rows [ys, ye) x columns [xs, xe) clipped to the image; empty -> {W,-1,H,-1} like dim_mask_bbox
    const int x0 = max(xs, 0), x1 = min(xe, W) - 1, y0 = max(ys, 0), y1 = min(ye, H) - 1;
    const bool any = bb[1] >= 0 && x1 >= x0 && y1 >= y0;
    int* o = bbox_of_mask + 4 * b;
    o[0] = any ? x0 : W; o[1] = any ? x1 : -1; o[2] = any ? y0 : H; o[3] = any ? y1 : -1;
  }
  bool row = (bb[1] >= 0) && y >= ys && y < ye;
  float4 v;
  v.x = (row && x4 + 0 >= xs && x4 + 0 < xe) ? 1.f : 0.f;
  v.y = (row && x4 + 1 >= xs && x4 + 1 < xe) ? 1.f : 0.f;
  v.z = (row && x4 + 2 >= xs && x4 + 2 < xe) ? 1.f : 0.f;
  v.w = (row && x4 + 3 >= xs && x4 + 3 < xe) ? 1.f : 0.f;
  *reinterpret_cast<float4*>(mask + ((long)b * H + y) * W + x4) = v;
}

// tester.py:221-225: light = 0.5 * dir, then += tx, -= ty, -= tz (float64 on the host there; the uniform is float32)
__global__ void modelnet_light_kernel(const float* __restrict__ poses, float dx, float dy, float dz, float* __restrict__ light_pos, int B) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float* P = poses + 12 * b;
  light_pos[3 * b + 0] = (float)(0.5 * (double)dx + (double)P[3]);
  light_pos[3 * b + 1] = (float)(0.5 * (double)dy - (double)P[7]);
  light_pos[3 * b + 2] = (float)(0.5 * (double)dz - (double)P[11]);
}

}  // namespace dim

using namespace dim;

extern "C" {

// workspace layout: header (256 B, ResolveHdr) | z-buffer (u64 per pixel) | projected vertices (3 floats each, padded to 256 B) |
// covered-pixel list | wave extents (one int4 per wave of the stream pass)
static long raster_scr_bytes(int B, int vmax) { return ((long)B * vmax * 3 * 4 + 255) / 256 * 256; }
static int raster_waves_per_sample(int H, int W) { return (int)(ceil_div((long)H * W / 4, 256) * 4); }   // workgroups of 256 quads x 4 waves

long dim_raster_workspace_bytes(int B, int vmax, int H, int W) {
  return (long)B * H * W * 8 + raster_scr_bytes(B, vmax) + (long)sizeof(ResolveHdr) + (long)B * H * W * 4 +
         (long)B * raster_waves_per_sample(H, W) * 16;
}

static int raster_render_impl(const float* verts, const float* normals, const float* uvs, const int* faces, const int* mesh_table,
                              int n_classes, int vmax, int fmax, const unsigned char* textures, const int* tex_table,
                              const int* class_index, const float* poses, const float* K9, int B, int H, int W, float znear, float zfar,
                              int tex_bilinear, const float* light_pos, const float* light_int, float ratio, const float* plane_means3,
                              float mask_thr, void* workspace, float* image, float* depth, float* mask, float* bgr, int* bbox, int* status,
                              const int* clean_bbox, void* stream) {
  if (B == 0) return DIM_OK;  // empty batch: nothing to do, pointers may be NULL
  DIM_REQUIRE(verts && uvs && faces && mesh_table && textures && tex_table && class_index && poses && K9 && workspace, "null pointer");
  DIM_REQUIRE(vmax > 0 && fmax > 0 && H > 0 && W > 0 && n_classes > 0, "bad sizes");
  DIM_REQUIRE((long)B * H * W < (1L << 32), "B * H * W must fit 32 bits (covered-pixel list entries)");
  DIM_REQUIRE(!image || plane_means3, "image output needs plane_means3");
  DIM_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 7) == 0, "workspace must be 8-byte aligned");
  hipStream_t st = as_stream(stream);
  char* ws = reinterpret_cast<char*>(workspace);
  ResolveHdr* hdr = reinterpret_cast<ResolveHdr*>(ws);
  unsigned long long* zbuf = reinterpret_cast<unsigned long long*>(ws + sizeof(ResolveHdr));
  float* scr = reinterpret_cast<float*>(ws + sizeof(ResolveHdr) + (long)B * H * W * 8);
  unsigned* list = reinterpret_cast<unsigned*>(ws + sizeof(ResolveHdr) + (long)B * H * W * 8 + raster_scr_bytes(B, vmax));
  const long nkeys = (long)B * H * W;
  auto aligned16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  // the two-pass resolve moves float4 / 2 x u64: every plane pointer (and the workspace) must be 16-byte aligned; anything else
  // takes the one-thread-per-pixel kernel.  (Every argument check comes BEFORE the first launch: a render that stopped between the
  // triangle pass and the last pass would leave a dirty z-buffer behind a header that calls it clear.)
  const bool two_pass = W % 4 == 0 && aligned16(workspace) && aligned16(image) && aligned16(depth) && aligned16(mask) && aligned16(bgr);
  DIM_REQUIRE(!clean_bbox || two_pass, "clean_bbox needs the two-pass resolve: W %% 4 == 0 and 16-byte aligned planes and workspace");
  // No clear pass and no init launch (round 4): the z-buffer is clear when a render ends (the pass that consumes a key resets it) and
  // the header in front of it says so; a header that does not match this call makes the vertex pass clear the z-buffer first.  (A
  // clear by hipMemsetAsync was never an option: inside a captured hipGraph the memset node was seen overlapping the resolve pass.)
  hipLaunchKernelGGL(raster_vertex_kernel, dim3(ceil_div(vmax, 256), B), dim3(256), 0, st, verts, mesh_table, class_index, poses,
                     K9[0], K9[4], K9[2], K9[5], vmax, n_classes, status, scr, hdr, zbuf, B, H, W, bbox,
                     (int)(aligned16(workspace) && nkeys % 2 == 0));
  hipLaunchKernelGGL(raster_tri_kernel, dim3(ceil_div(fmax, 256), B), dim3(256), 0, st, faces, mesh_table, class_index, scr, verts, poses,
                     K9[0], K9[4], K9[2], K9[5], vmax, H, W, znear, zfar, n_classes, zbuf, hdr, B);
  float p0 = plane_means3 ? plane_means3[0] : 0.f, p1 = plane_means3 ? plane_means3[1] : 0.f, p2 = plane_means3 ? plane_means3[2] : 0.f;
  LitArgs lit = {verts, normals, poses, light_pos, light_int, ratio, K9[0], K9[4], K9[2], K9[5], fmaxf(znear, kZClipMin)};
  if (two_pass) {
    // pass 1 streams the z-buffer once and finishes depth / mask / background (only inside the caller's dirty box, if it names one) and
    // lists the covered pixels; pass 2 folds the bbox, colours the listed pixels on full waves and resets their keys
    const int wps = raster_waves_per_sample(H, W);
    int4* wave_ext = bbox ? reinterpret_cast<int4*>(reinterpret_cast<char*>(list) + (long)B * H * W * 4) : nullptr;
    hipLaunchKernelGGL(raster_resolve_stream_kernel, dim3(ceil_div((long)H * W / 4, 256), B), dim3(256), 0, st, zbuf, H, W, p0, p1, p2,
                       mask_thr, image, depth, mask, bgr, wave_ext, wps, hdr, list, clean_bbox);
    int grid = (int)(nkeys / 256 < 2048 ? (nkeys + 255) / 256 : 2048);
    if (grid < B) grid = B;
    if (!(image || bgr))
      hipLaunchKernelGGL((raster_resolve_shade_kernel<false, false>), dim3(grid), dim3(256), 0, st, lit, uvs, faces, mesh_table, textures,
                         tex_table, class_index, scr, zbuf, vmax, H, W, tex_bilinear, p0, p1, p2, image, bgr, status, hdr, list, wave_ext,
                         wps, B, bbox);
    else if (normals)
      hipLaunchKernelGGL((raster_resolve_shade_kernel<true, true>), dim3(grid), dim3(256), 0, st, lit, uvs, faces, mesh_table, textures,
                         tex_table, class_index, scr, zbuf, vmax, H, W, tex_bilinear, p0, p1, p2, image, bgr, status, hdr, list, wave_ext,
                         wps, B, bbox);
    else
      hipLaunchKernelGGL((raster_resolve_shade_kernel<false, true>), dim3(grid), dim3(256), 0, st, lit, uvs, faces, mesh_table, textures,
                         tex_table, class_index, scr, zbuf, vmax, H, W, tex_bilinear, p0, p1, p2, image, bgr, status, hdr, list, wave_ext,
                         wps, B, bbox);
  } else {
    if (normals)
      hipLaunchKernelGGL(raster_resolve_kernel<true>, dim3(ceil_div(W, 256), H, B), dim3(256), 0, st, lit, uvs, faces, mesh_table, textures,
                         tex_table, class_index, scr, zbuf, vmax, H, W, tex_bilinear, p0, p1, p2, mask_thr, image, depth, mask, bgr, bbox,
                         status);
    else
      hipLaunchKernelGGL(raster_resolve_kernel<false>, dim3(ceil_div(W, 256), H, B), dim3(256), 0, st, lit, uvs, faces, mesh_table,
                         textures, tex_table, class_index, scr, zbuf, vmax, H, W, tex_bilinear, p0, p1, p2, mask_thr, image, depth, mask,
                         bgr, bbox, status);
  }
  return check_launch("raster_render");
}

int dim_raster_render(const float* verts, const float* uvs, const int* faces, const int* mesh_table, int n_classes, int vmax, int fmax,
                      const unsigned char* textures, const int* tex_table, const int* class_index, const float* poses,
                      const float* K9, int B, int H, int W, float znear, float zfar, int tex_bilinear, const float* plane_means3,
                      float mask_thr, void* workspace, float* image, float* depth, float* mask, float* bgr, int* bbox, int* status,
                      void* stream) {
  return raster_render_impl(verts, nullptr, uvs, faces, mesh_table, n_classes, vmax, fmax, textures, tex_table, class_index, poses, K9, B,
                            H, W, znear, zfar, tex_bilinear, nullptr, nullptr, 0.f, plane_means3, mask_thr, workspace, image, depth, mask,
                            bgr, bbox, status, nullptr, stream);
}

int dim_raster_render_lit(const float* verts, const float* normals, const float* uvs, const int* faces, const int* mesh_table,
                          int n_classes, int vmax, int fmax, const unsigned char* textures, const int* tex_table, const int* class_index,
                          const float* poses, const float* K9, int B, int H, int W, float znear, float zfar, int tex_bilinear,
                          const float* light_pos, const float* light_int, float brightness_ratio, const float* plane_means3,
                          float mask_thr, void* workspace, float* image, float* depth, float* mask, float* bgr, int* bbox, int* status,
                          void* stream) {
  if (B == 0) return DIM_OK;
  DIM_REQUIRE(normals && light_pos && light_int, "null pointer");
  return raster_render_impl(verts, normals, uvs, faces, mesh_table, n_classes, vmax, fmax, textures, tex_table, class_index, poses, K9, B,
                            H, W, znear, zfar, tex_bilinear, light_pos, light_int, brightness_ratio, plane_means3, mask_thr, workspace,
                            image, depth, mask, bgr, bbox, status, nullptr, stream);
}

int dim_raster_render_dirty(const float* verts, const float* normals, const float* uvs, const int* faces, const int* mesh_table,
                            int n_classes, int vmax, int fmax, const unsigned char* textures, const int* tex_table, const int* class_index,
                            const float* poses, const float* K9, int B, int H, int W, float znear, float zfar, int tex_bilinear,
                            const float* light_pos, const float* light_int, float brightness_ratio, const float* plane_means3,
                            float mask_thr, void* workspace, float* image, float* depth, float* mask, float* bgr, int* bbox, int* status,
                            const int* clean_bbox, void* stream) {
  if (B == 0) return DIM_OK;
  DIM_REQUIRE(!normals || (light_pos && light_int), "lit render: null light pointer");
  DIM_REQUIRE(!clean_bbox || clean_bbox != bbox, "clean_bbox and bbox must be different arrays (the stream pass reads one while the last pass writes the other)");
  return raster_render_impl(verts, normals, uvs, faces, mesh_table, n_classes, vmax, fmax, textures, tex_table, class_index, poses, K9, B,
                            H, W, znear, zfar, tex_bilinear, light_pos, light_int, brightness_ratio, plane_means3, mask_thr, workspace,
                            image, depth, mask, bgr, bbox, status, clean_bbox, stream);
}

int dim_modelnet_light_position(const float* poses, float dx, float dy, float dz, float* light_pos, int B, void* stream) {
  if (B == 0) return DIM_OK;
  DIM_REQUIRE(poses && light_pos, "null pointer");
  hipLaunchKernelGGL(modelnet_light_kernel, dim3(ceil_div(B, 64)), dim3(64), 0, as_stream(stream), poses, dx, dy, dz, light_pos, B);
  return check_launch("modelnet_light_position");
}

int dim_box_mask(const int* bbox, float* mask, int B, int H, int W, int* bbox_of_mask, void* stream) {
  if (B == 0) return DIM_OK;  // empty batch: nothing to do, pointers may be NULL
  DIM_REQUIRE(bbox && mask, "null pointer");
  DIM_REQUIRE(W % 4 == 0, "W must be a multiple of 4");
  hipLaunchKernelGGL(box_mask_kernel, dim3(ceil_div(W / 4, 256), H, B), dim3(256), 0, as_stream(stream), bbox, mask, H, W,
                     bbox_of_mask);
  return check_launch("box_mask");
}

}  // extern "C"
