/* Software rasteriser -- CPU oracle for lib/render_glumpy/render_py_multi.py (test-only).
 *
 * Restates what Render_Py.render (render_py_multi.py:112-147) asks OpenGL to do:
 *   - view = diag(1,-1,-1)*[R|t] (:171-178) and the pinhole projection of
 *     my_compute_calib_proj (:152-169).  Deriving window coordinates from that matrix
 *     gives  x_win = fx*X/Z + cx + 0.5,  y_win(top-down after flipud) = fy*Y/Z + cy + 0.5,
 *     i.e. the centre of pixel (i,j) samples the projection at (u,v) = (i,j).
 *   - GL depth buffer -> metric depth (:139-146) is exactly camera-frame Z; background 0.
 *   - unlit texture lookup (fragment shader :36-46) of texture_map.png uploaded flipped
 *     vertically (:76-78); colour returned as BGR * 255 (:135-137).
 * OpenGL's fill rule / sub-pixel snapping / texture filter are implementation-defined
 * (glumpy + driver absent here): this oracle fixes them as 8-bit sub-pixel snapping,
 * top-left fill rule, nearest (default) or bilinear clamp-to-edge filtering.
 * "parity unpinned" w.r.t. a real GL context; the HIP rasteriser is checked against THIS.
 *
 * Lit variant (lib/render_glumpy/render_py_light_modelnet_multi.py): the fragment shader (:36-77) shades
 * the texel with  (1-ratio) + ratio*clamp(cos(normal, light - position), 0, 1)  times the light
 * intensity, in GL camera coordinates (view = diag(1,-1,-1)*[R|t], :256-262; the normal matrix :184-196
 * is the inverse transpose of the view, i.e. its rotation part for a rigid view, and the shader divides by
 * length(normal) again so the vec4 normalisation cancels).  The 8-bit framebuffer quantises the clamped colour
 * with round-to-nearest (:211-214 reads it back as float and rounds again, a no-op).
 *
 * Near plane (GL clips primitives against zNear, render_py_multi.py:152-169): a triangle whose three vertices lie at
 * Z >= zNear is rasterised as it is (fragments outside [znear, zfar] are discarded per pixel, which for such a triangle is
 * what geometric clipping gives); one with all three in front of the plane is dropped; one that STRADDLES the plane is
 * clipped in camera space (Sutherland-Hodgman against Z = zNear, intersection always computed from the inside to the outside
 * vertex so that two triangles sharing the edge cut it in the same point), the 3- or 4-gon is rasterised as a fan, and its
 * fragments are shaded with barycentrics of the ORIGINAL triangle taken in camera space (its screen-space triangle does not
 * exist when a vertex is behind the eye).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define SUBPIX 256.0f
#define COORD_LIM 1.0e6f

typedef struct {
  int64_t A[3], B[3], C[3];
  int64_t area;
  int tl[3];
} edges_t;

static inline int32_t snap(float u) { return (int32_t)floorf(u * SUBPIX + 0.5f); }

/* returns 0 when degenerate */
static int setup_edges(const int32_t X[3], const int32_t Y[3], edges_t *e) {
  /* edge i is opposite vertex i: from v[(i+1)%3] to v[(i+2)%3] */
  for (int i = 0; i < 3; ++i) {
    int a = (i + 1) % 3, b = (i + 2) % 3;
    e->A[i] = (int64_t)Y[a] - (int64_t)Y[b];
    e->B[i] = (int64_t)X[b] - (int64_t)X[a];
    e->C[i] = (int64_t)X[a] * (int64_t)Y[b] - (int64_t)X[b] * (int64_t)Y[a];
  }
  e->area = e->A[0] * X[0] + e->B[0] * Y[0] + e->C[0];
  if (e->area == 0) return 0;
  if (e->area < 0) {
    for (int i = 0; i < 3; ++i) { e->A[i] = -e->A[i]; e->B[i] = -e->B[i]; e->C[i] = -e->C[i]; }
    e->area = -e->area;
  }
  for (int i = 0; i < 3; ++i) e->tl[i] = (e->A[i] > 0) || (e->A[i] == 0 && e->B[i] > 0);
  return 1;
}

static inline int inside(const edges_t *e, int64_t px, int64_t py, int64_t E[3]) {
  for (int i = 0; i < 3; ++i) {
    E[i] = e->A[i] * px + e->B[i] * py + e->C[i];
    if (E[i] < 0 || (E[i] == 0 && !e->tl[i])) return 0;
  }
  return 1;
}

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* One view. verts (V,3), uvs (V,2), faces (F,3) int32; tex (Ht,Wt,3) uint8 RGB, row 0 = top of the
 * image file; R (9) row-major, t (3), K (9); outputs bgr (H,W,3) float 0..255, depth (H,W) float.
 * shade: NULL, or per-vertex intensity (V) multiplied into the colour then rounded (ModelNet variant).
 * scratch: caller-provided H*W uint64 z-buffer + V*3 floats (cam u,v,z). */
typedef struct {
  const float *normals;   /* (V,3) per-vertex normals, model frame */
  const float *light_pos; /* (3) GL camera coordinates (tester.py:221-225) */
  const float *light_int; /* (3) rgb intensity */
  float ratio;            /* brightness_ratio (0.7 in tester.py:190) */
} lit_t;

#define ZCLIP_MIN 1.0e-4f

/* Sutherland-Hodgman against z >= zc; cam = the triangle in camera space; returns 0, 3 or 4 vertices in out */
static int clip_near(float cam[3][3], float zc, float out[4][3]) {
  int n = 0;
  for (int i = 0; i < 3; ++i) {
    const float *a = cam[i], *b = cam[(i + 1) % 3];
    const int ina = a[2] >= zc, inb = b[2] >= zc;
    if (ina) { out[n][0] = a[0]; out[n][1] = a[1]; out[n][2] = a[2]; ++n; }
    if (ina != inb) {
      const float *pi = ina ? a : b, *po = ina ? b : a;
      const float tt = (zc - pi[2]) / (po[2] - pi[2]);
      out[n][0] = fmaf(tt, po[0] - pi[0], pi[0]);
      out[n][1] = fmaf(tt, po[1] - pi[1], pi[1]);
      out[n][2] = zc;
      ++n;
    }
  }
  return n;
}

/* barycentrics (affine, camera space) of the point that pixel (x, y) sees at depth z, w.r.t. the triangle cam; float64 */
static void cam_bary(float cam[3][3], int x, int y, float z, float fx, float fy, float cx, float cy, float w[3]) {
  const double P[3] = {(double)z * (((double)x - (double)cx) / (double)fx), (double)z * (((double)y - (double)cy) / (double)fy), (double)z};
  double e1[3], e2[3], d0[3], d1[3], d2[3];
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

/* one screen triangle into the z-buffer; clamp_near: fragments of a clipped piece may round below the plane they were cut at */
static void raster_one(const int32_t X[3], const int32_t Y[3], const float iz[3], int f, int H, int W, float znear, float zfar,
                       int clamp_near, float zc, uint64_t *zbuf) {
  edges_t e;
  if (!setup_edges(X, Y, &e)) return;
  int32_t minX = X[0] < X[1] ? X[0] : X[1]; if (X[2] < minX) minX = X[2];
  int32_t maxX = X[0] > X[1] ? X[0] : X[1]; if (X[2] > maxX) maxX = X[2];
  int32_t minY = Y[0] < Y[1] ? Y[0] : Y[1]; if (Y[2] < minY) minY = Y[2];
  int32_t maxY = Y[0] > Y[1] ? Y[0] : Y[1]; if (Y[2] > maxY) maxY = Y[2];
  int x0 = (minX + 255) >> 8, x1 = maxX >> 8, y0 = (minY + 255) >> 8, y1 = maxY >> 8; /* arithmetic shifts = floor */
  if (x0 < 0) x0 = 0; if (y0 < 0) y0 = 0; if (x1 > W - 1) x1 = W - 1; if (y1 > H - 1) y1 = H - 1;
  const float inv_area = 1.0f / (float)e.area;
  for (int y = y0; y <= y1; ++y)
    for (int x = x0; x <= x1; ++x) {
      int64_t E[3];
      if (!inside(&e, (int64_t)x * 256, (int64_t)y * 256, E)) continue;
      float b0 = (float)E[0] * inv_area, b1 = (float)E[1] * inv_area, b2 = (float)E[2] * inv_area;
      float invz = fmaf(b2, iz[2], fmaf(b1, iz[1], b0 * iz[0]));
      float z = 1.0f / invz;
      if (clamp_near && z < zc) z = zc;
      if (!(z >= znear && z <= zfar)) continue;
      uint64_t key = ((uint64_t)f2u(z) << 32) | (uint32_t)f;
      uint64_t *zp = zbuf + (size_t)y * W + x;
      if (key < *zp) *zp = key;
    }
}

static void render_impl(const float *verts, const float *uvs, const int32_t *faces, int V, int F,
                        const uint8_t *tex, int Ht, int Wt, const float *R, const float *t, const float *K,
                        int H, int W, float znear, float zfar, int tex_bilinear, const lit_t *lit, float *bgr, float *depth) {
  uint64_t *zbuf = (uint64_t *)malloc((size_t)H * W * sizeof(uint64_t));
  float *scr = (float *)malloc((size_t)V * 3 * sizeof(float));
  memset(zbuf, 0xFF, (size_t)H * W * sizeof(uint64_t));
  const float fx = K[0], cx = K[2], fy = K[4], cy = K[5];
  for (int i = 0; i < V; ++i) {
    const float *p = verts + 3 * i;
    float xc = fmaf(R[0], p[0], fmaf(R[1], p[1], fmaf(R[2], p[2], t[0])));
    float yc = fmaf(R[3], p[0], fmaf(R[4], p[1], fmaf(R[5], p[2], t[1])));
    float zc = fmaf(R[6], p[0], fmaf(R[7], p[1], fmaf(R[8], p[2], t[2])));
    scr[3 * i + 0] = fmaf(fx, xc / zc, cx);
    scr[3 * i + 1] = fmaf(fy, yc / zc, cy);
    scr[3 * i + 2] = zc;
  }
  const float zc = znear > ZCLIP_MIN ? znear : ZCLIP_MIN;
  for (int f = 0; f < F; ++f) {
    int32_t X[3], Y[3];
    float iz[3];
    const float z0 = scr[3 * faces[3 * f] + 2], z1 = scr[3 * faces[3 * f + 1] + 2], z2 = scr[3 * faces[3 * f + 2] + 2];
    const int nin = (z0 >= zc) + (z1 >= zc) + (z2 >= zc);
    if (nin == 0) continue; /* wholly in front of the near plane (or NaN): clipped away */
    if (nin < 3) {          /* straddles the near plane: clip in camera space, rasterise the pieces */
      float cam[3][3], poly[4][3];
      for (int k = 0; k < 3; ++k) {
        const float *p = verts + 3 * faces[3 * f + k];
        cam[k][0] = fmaf(R[0], p[0], fmaf(R[1], p[1], fmaf(R[2], p[2], t[0])));
        cam[k][1] = fmaf(R[3], p[0], fmaf(R[4], p[1], fmaf(R[5], p[2], t[1])));
        cam[k][2] = fmaf(R[6], p[0], fmaf(R[7], p[1], fmaf(R[8], p[2], t[2])));
      }
      const int np = clip_near(cam, zc, poly);
      float su[4], sv[4];
      int ok = np >= 3;
      for (int k = 0; k < np; ++k) {
        su[k] = fmaf(fx, poly[k][0] / poly[k][2], cx);
        sv[k] = fmaf(fy, poly[k][1] / poly[k][2], cy);
        if (!(fabsf(su[k]) < COORD_LIM) || !(fabsf(sv[k]) < COORD_LIM)) ok = 0;
      }
      if (!ok) continue;
      for (int piece = 0; piece + 2 < np; ++piece) {
        const int idx[3] = {0, piece + 1, piece + 2};
        for (int k = 0; k < 3; ++k) { X[k] = snap(su[idx[k]]); Y[k] = snap(sv[idx[k]]); iz[k] = 1.0f / poly[idx[k]][2]; }
        raster_one(X, Y, iz, f, H, W, znear, zfar, 1, zc, zbuf);
      }
      continue;
    }
    int ok = 1;
    for (int k = 0; k < 3; ++k) {
      const float *s = scr + 3 * faces[3 * f + k];
      if (!(fabsf(s[0]) < COORD_LIM) || !(fabsf(s[1]) < COORD_LIM)) { ok = 0; break; }
      X[k] = snap(s[0]); Y[k] = snap(s[1]); iz[k] = 1.0f / s[2];
    }
    if (!ok) continue;
    raster_one(X, Y, iz, f, H, W, znear, zfar, 0, zc, zbuf);
  }
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
      uint64_t key = zbuf[(size_t)y * W + x];
      float *o = bgr + ((size_t)y * W + x) * 3;
      if (key == UINT64_MAX) { o[0] = o[1] = o[2] = 0.f; depth[(size_t)y * W + x] = 0.f; continue; }
      int f = (int)(uint32_t)(key & 0xFFFFFFFFu);
      float z = u2f((uint32_t)(key >> 32));
      int32_t X[3], Y[3];
      float iz[3], tu[3], tv[3];
      int clipped = 0;
      for (int k = 0; k < 3; ++k) {
        int vi = faces[3 * f + k];
        tu[k] = uvs[2 * vi]; tv[k] = uvs[2 * vi + 1];
        if (!(scr[3 * vi + 2] >= zc)) clipped = 1;
      }
      float w0, w1, w2, zs; /* attribute = (w2 a2 + w1 a1 + w0 a0) * zs */
      if (clipped) {        /* a piece of a near-clipped triangle: camera-space barycentrics of the original */
        float cam[3][3], wb[3];
        for (int k = 0; k < 3; ++k) {
          const float *p = verts + 3 * faces[3 * f + k];
          cam[k][0] = fmaf(R[0], p[0], fmaf(R[1], p[1], fmaf(R[2], p[2], t[0])));
          cam[k][1] = fmaf(R[3], p[0], fmaf(R[4], p[1], fmaf(R[5], p[2], t[1])));
          cam[k][2] = fmaf(R[6], p[0], fmaf(R[7], p[1], fmaf(R[8], p[2], t[2])));
        }
        cam_bary(cam, x, y, z, fx, fy, cx, cy, wb);
        w0 = wb[0]; w1 = wb[1]; w2 = wb[2]; zs = 1.0f;
      } else {
        for (int k = 0; k < 3; ++k) {
          const float *s = scr + 3 * faces[3 * f + k];
          X[k] = snap(s[0]); Y[k] = snap(s[1]); iz[k] = 1.0f / s[2];
        }
        edges_t e; int64_t E[3];
        setup_edges(X, Y, &e);
        inside(&e, (int64_t)x * 256, (int64_t)y * 256, E);
        const float inv_area = 1.0f / (float)e.area;
        float b0 = (float)E[0] * inv_area, b1 = (float)E[1] * inv_area, b2 = (float)E[2] * inv_area;
        w0 = b0 * iz[0]; w1 = b1 * iz[1]; w2 = b2 * iz[2]; zs = z;
      }
      float u = fmaf(w2, tu[2], fmaf(w1, tu[1], w0 * tu[0])) * zs;
      float v = fmaf(w2, tv[2], fmaf(w1, tv[1], w0 * tv[0])) * zs;
      float rgb[3];
      if (!tex_bilinear) {
        int tx = clampi((int)floorf(u * (float)Wt), 0, Wt - 1);
        int ty = clampi((int)floorf(v * (float)Ht), 0, Ht - 1);
        const uint8_t *px = tex + ((size_t)(Ht - 1 - ty) * Wt + tx) * 3;
        rgb[0] = px[0]; rgb[1] = px[1]; rgb[2] = px[2];
      } else {
        float xf = u * (float)Wt - 0.5f, yf = v * (float)Ht - 0.5f;
        float x0f = floorf(xf), y0f = floorf(yf);
        float ax = xf - x0f, ay = yf - y0f;
        int xa = clampi((int)x0f, 0, Wt - 1), xb = clampi((int)x0f + 1, 0, Wt - 1);
        int ya = clampi((int)y0f, 0, Ht - 1), yb = clampi((int)y0f + 1, 0, Ht - 1);
        const uint8_t *p00 = tex + ((size_t)(Ht - 1 - ya) * Wt + xa) * 3, *p01 = tex + ((size_t)(Ht - 1 - ya) * Wt + xb) * 3;
        const uint8_t *p10 = tex + ((size_t)(Ht - 1 - yb) * Wt + xa) * 3, *p11 = tex + ((size_t)(Ht - 1 - yb) * Wt + xb) * 3;
        for (int c = 0; c < 3; ++c) {
          float top = fmaf(ax, (float)p01[c] - (float)p00[c], (float)p00[c]);
          float bot = fmaf(ax, (float)p11[c] - (float)p10[c], (float)p10[c]);
          rgb[c] = fmaf(ay, bot - top, top);
          if (!lit) rgb[c] = floorf(rgb[c]); /* tester.py:244 astype('uint8') truncation */
        }
      }
      if (lit) {
        float n[3], p[3];
        for (int c = 0; c < 3; ++c) {
          const float *N0 = lit->normals + 3 * faces[3 * f], *N1 = lit->normals + 3 * faces[3 * f + 1], *N2 = lit->normals + 3 * faces[3 * f + 2];
          const float *P0 = verts + 3 * faces[3 * f], *P1 = verts + 3 * faces[3 * f + 1], *P2 = verts + 3 * faces[3 * f + 2];
          n[c] = fmaf(w2, N2[c], fmaf(w1, N1[c], w0 * N0[c])) * zs; /* perspective-correct varyings v_normal, v_position */
          p[c] = fmaf(w2, P2[c], fmaf(w1, P1[c], w0 * P0[c])) * zs;
        }
        float Ng[3], Pg[3];
        for (int r = 0; r < 3; ++r) {
          float sgn = r == 0 ? 1.f : -1.f; /* OpenCV -> OpenGL camera: y and z flip */
          Ng[r] = sgn * fmaf(R[3 * r + 2], n[2], fmaf(R[3 * r + 1], n[1], R[3 * r] * n[0]));
          Pg[r] = sgn * (fmaf(R[3 * r + 2], p[2], fmaf(R[3 * r + 1], p[1], R[3 * r] * p[0])) + t[r]);
        }
        float sx = lit->light_pos[0] - Pg[0], sy = lit->light_pos[1] - Pg[1], sz = lit->light_pos[2] - Pg[2];
        float dotv = fmaf(Ng[2], sz, fmaf(Ng[1], sy, Ng[0] * sx));
        float ls = sqrtf(fmaf(sz, sz, fmaf(sy, sy, sx * sx)));
        float ln = sqrtf(fmaf(Ng[2], Ng[2], fmaf(Ng[1], Ng[1], Ng[0] * Ng[0])));
        float br = dotv / (ls * ln);
        br = fmaxf(fminf(br, 1.0f), 0.0f);
        float k = fmaf(lit->ratio, br, 1.0f - lit->ratio);
        for (int c = 0; c < 3; ++c) {
          float col = (rgb[c] / 255.0f) * (k * lit->light_int[c]);
          col = fminf(fmaxf(col, 0.0f), 1.0f);
          rgb[c] = floorf(fmaf(col, 255.0f, 0.5f));
        }
      }
      o[0] = rgb[2]; o[1] = rgb[1]; o[2] = rgb[0];
      depth[(size_t)y * W + x] = z;
    }
  free(zbuf);
  free(scr);
}

void dim_oracle_render(const float *verts, const float *uvs, const int32_t *faces, int V, int F,
                       const uint8_t *tex, int Ht, int Wt, const float *R, const float *t, const float *K,
                       int H, int W, float znear, float zfar, int tex_bilinear, float *bgr, float *depth) {
  render_impl(verts, uvs, faces, V, F, tex, Ht, Wt, R, t, K, H, W, znear, zfar, tex_bilinear, NULL, bgr, depth);
}

/* Render_Py_Light_ModelNet_Multi.render (render_py_light_modelnet_multi.py:153-235) */
void dim_oracle_render_lit(const float *verts, const float *normals, const float *uvs, const int32_t *faces, int V, int F,
                           const uint8_t *tex, int Ht, int Wt, const float *R, const float *t, const float *K,
                           int H, int W, float znear, float zfar, int tex_bilinear, const float *light_pos,
                           const float *light_int, float ratio, float *bgr, float *depth) {
  lit_t lit = {normals, light_pos, light_int, ratio};
  render_impl(verts, uvs, faces, V, F, tex, Ht, Wt, R, t, K, H, W, znear, zfar, tex_bilinear, &lit, bgr, depth);
}

/* gpu_flow_kernel.cu:32-69 restated on the CPU (float arithmetic, same operation order). */
void dim_oracle_flow(const float *depth_src, const float *depth_tgt, const float *KT, const float *Kinv,
                     int B, int H, int W, float *flow, float *valid) {
  for (int b = 0; b < B; ++b)
    for (int h = 0; h < H; ++h)
      for (int w = 0; w < W; ++w) {
        size_t index = ((size_t)b * H + h) * W + w;
        const float *kt = KT + 12 * b;
        float d = depth_src[index];
        float x = (w * Kinv[0] + h * Kinv[1] + Kinv[2]) * d;
        float y = (w * Kinv[3] + h * Kinv[4] + Kinv[5]) * d;
        float z = d;
        float fh = 0.f, fw = 0.f, va = 0.f;
        if (d > 1E-3) {
          float xp = x * kt[0] + y * kt[1] + z * kt[2] + kt[3];
          float yp = x * kt[4] + y * kt[5] + z * kt[6] + kt[7];
          float zp = (float)((double)(x * kt[8] + y * kt[9] + z * kt[10] + kt[11]) + 1E-15);
          float wp = xp / zp, hp = yp / zp;
          int wi = (int)round((double)wp), hi = (int)round((double)hp);
          if (wp >= 0 && wp <= W - 1 && hp >= 0 && hp <= H - 1) {
            float dt = depth_tgt[((size_t)b * H + hi) * W + wi];
            if (fabsf(zp - dt) < 3E-3) { fh = hp - h; fw = wp - w; va = 1.f; }
          }
        }
        flow[(((size_t)b * 2 + 0) * H + h) * W + w] = fh;
        flow[(((size_t)b * 2 + 1) * H + h) * W + w] = fw;
        valid[index] = va;
      }
}
