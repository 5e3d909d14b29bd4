// k_atlas_upload.hip -- images on their way into the atlas: the LCD filter of glyph images (common/textrasters/pixie_raster.nim:12-43),
// pixie's minifyBy2 for the mip chain (opengl/textures.nim:106-119), the blit into a level, the coverage rasteriser of glyph outlines,
// and a fill.
#include "fdh_device.h"

namespace fdh {
// ------------------------------------------------------------------ glyph images on their way into the atlas
// FreeType's default 5-tap LCD filter as the reference applies it to a rasterised glyph before the upload
// (common/textrasters/pixie_raster.nim:12-43): weights 8, 77, 86, 77, 8 over x - 2 .. x + 2 with the column clamped to the image,
// per channel (sum + 128) >> 8.  Integer arithmetic: bit-exact with the oracle's restatement.
__global__ void k_lcd_filter(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, int w, int h) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= w || y >= h) return;
  const int wt[5] = {8, 77, 86, 77, 8};
  int sr = 0, sg = 0, sb = 0, sa = 0;
#pragma unroll
  for (int i = 0; i < 5; i++) {
    const int sx = min(max(x + i - 2, 0), w - 1);
    const uint32_t p = src[(size_t)y * w + sx];
    sr += (int)(p & 255u) * wt[i]; sg += (int)((p >> 8) & 255u) * wt[i]; sb += (int)((p >> 16) & 255u) * wt[i]; sa += (int)(p >> 24) * wt[i];
  }
  dst[(size_t)y * w + x] = (uint32_t)(((sr + 128) >> 8) & 255) | ((uint32_t)(((sg + 128) >> 8) & 255) << 8) |
                           ((uint32_t)(((sb + 128) >> 8) & 255) << 16) | ((uint32_t)(((sa + 128) >> 8) & 255) << 24);
}
// one mip step of updateSubImage (textures.nim:106-119) = pixie's Image.minifyBy2 on premultiplied RGBA8: box sum div 4; an odd
// extent rounds the result size up and the extra column / row / corner carry half / half / quarter coverage (the arithmetic
// is pinned by the reference's data/img1.flippy: minify_by2_host in fdh_context.cpp spells it out)
__global__ void k_minify2(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, int sw, int sh) {
  const int nw = (sw + 1) >> 1, nh = (sh + 1) >> 1;
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= nw || y >= nh) return;
  const bool col_pair = 2 * x + 1 < sw, row_pair = 2 * y + 1 < sh;
  const int x0 = col_pair ? 2 * x : sw - 1, x1 = col_pair ? 2 * x + 1 : sw - 1, y0 = row_pair ? 2 * y : sh - 1, y1 = row_pair ? 2 * y + 1 : sh - 1;
  const uint32_t a = src[(size_t)y0 * sw + x0], b = src[(size_t)y0 * sw + x1], c = src[(size_t)y1 * sw + x0], d = src[(size_t)y1 * sw + x1];
  uint32_t o = 0;
#pragma unroll
  for (int k = 0; k < 32; k += 8) {
    const uint32_t ca = (a >> k) & 255u, cb = (b >> k) & 255u, cc = (c >> k) & 255u, cd = (d >> k) & 255u;
    uint32_t v;
    if (col_pair && row_pair) v = (ca + cb + cc + cd) >> 2;
    else if (row_pair) v = ((ca * 127u + cc * 128u) / 255u) * 128u / 255u;  // last column: rows 2y, 2y + 1
    else if (col_pair) v = ((ca * 127u + cb * 128u) / 255u) * 128u / 255u;  // last row: columns 2x, 2x + 1
    else v = ca * 64u / 255u;
    o |= v << k;
  }
  dst[(size_t)y * nw + x] = o;
}
// a w x h image into the rectangle (x, y) of one atlas level (LS texels wide); texels outside the level are dropped
__global__ void k_atlas_blit(uint32_t* __restrict__ level, int LS, int x, int y, const uint32_t* __restrict__ src, int w, int h) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y;
  if (i >= w || j >= h) return;
  const int tx = x + i, ty = y + j;
  if (tx >= 0 && ty >= 0 && tx < LS && ty < LS) level[(size_t)ty * LS + tx] = src[(size_t)j * w + i];
}
// Glyph outline -> coverage: exact-area scanline accumulation (oracle/figdraw_oracle.c, raster_row_line, operation for operation:
// no FMA contraction here so that both produce the same floats).  One lane owns one pixel row: it walks every line segment in
// order and adds the signed areas of the part inside its row to the row's accumulation cells (global scratch, w + 2 floats per
// row: nobody else touches them), then a running sum along the row turns areas into coverage.  Out: premultiplied white.
__device__ __forceinline__ void raster_row_line(float* acc, int w, int y, float x0, float y0, float x1, float y1) {
#pragma clang fp contract(off)
  if (y0 == y1) return;
  float dir = 1.0f;
  if (y0 > y1) { float t = x0; x0 = x1; x1 = t; t = y0; y0 = y1; y1 = t; dir = -1.0f; }
  const float ya = y0 > (float)y ? y0 : (float)y, yb = y1 < (float)(y + 1) ? y1 : (float)(y + 1);
  if (!(yb > ya)) return;
  const float dxdy = (x1 - x0) / (y1 - y0);
  const float xa = x0 + (ya - y0) * dxdy, xb = x0 + (yb - y0) * dxdy;
  const float d = (yb - ya) * dir;
  float xl = xa < xb ? xa : xb, xr = xa < xb ? xb : xa;
  if (xl < 0.0f) xl = 0.0f;
  if (xr < 0.0f) xr = 0.0f;
  if (xl > (float)w) xl = (float)w;
  if (xr > (float)w) xr = (float)w;
  const float x0floor = __builtin_floorf(xl);
  const int x0i = (int)x0floor;
  const float x1ceil = __builtin_ceilf(xr);
  const int x1i = (int)x1ceil;
  if (x1i <= x0i + 1) {
    const float xmf = 0.5f * (xl + xr) - x0floor;
    acc[x0i] += d - d * xmf;
    if (x0i + 1 <= w) acc[x0i + 1] += d * xmf;
  } else {
    const float s = 1.0f / (xr - xl);
    const float x0f = xl - x0floor;
    const float a0 = 0.5f * s * (1.0f - x0f) * (1.0f - x0f);
    const float x1f = xr - x1ceil + 1.0f;
    const float am = 0.5f * s * x1f * x1f;
    acc[x0i] += d * a0;
    if (x1i == x0i + 2) {
      acc[x0i + 1] += d * (1.0f - a0 - am);
    } else {
      const float a1 = s * (1.5f - x0f);
      acc[x0i + 1] += d * (a1 - a0);
      for (int xi = x0i + 2; xi < x1i - 1; xi++) acc[xi] += d * s;
      const float a2 = a1 + (float)(x1i - x0i - 3) * s;
      acc[x1i - 1] += d * (1.0f - a2 - am);
    }
    if (x1i <= w) acc[x1i] += d * am;
  }
}
__global__ __launch_bounds__(64) void k_rasterize_lines(const float4* __restrict__ lines, int n, int w, int h, float* __restrict__ scratch, uint32_t* __restrict__ out) {
#pragma clang fp contract(off)
  const int y = blockIdx.x * 64 + threadIdx.x;
  if (y >= h) return;
  float* acc = scratch + (size_t)y * (w + 2);
  for (int x = 0; x < w + 2; x++) acc[x] = 0.0f;
  for (int i = 0; i < n; i++) {
    const float4 l = lines[i];
    raster_row_line(acc, w, y, l.x, l.y, l.z, l.w);
  }
  float sum = 0.0f;
  for (int x = 0; x < w; x++) {
    sum += acc[x];
    float c = __builtin_fabsf(sum);
    if (c > 1.0f) c = 1.0f;
    const uint32_t v = (uint32_t)(c * 255.0f + 0.5f);
    out[(size_t)y * w + x] = v * 0x01010101u;
  }
}
void launch_rasterize_lines(hipStream_t s, const float4* lines, int n, int w, int h, float* scratch, uint32_t* out) {
  if (w > 0 && h > 0) hipLaunchKernelGGL(k_rasterize_lines, dim3((h + 63) / 64), dim3(64), 0, s, lines, n, w, h, scratch, out);
}
void launch_lcd_filter(hipStream_t s, const uint32_t* src, uint32_t* dst, int w, int h) {
  if (w > 0 && h > 0) hipLaunchKernelGGL(k_lcd_filter, dim3((w + 63) / 64, h), dim3(64), 0, s, src, dst, w, h);
}
void launch_minify2(hipStream_t s, const uint32_t* src, uint32_t* dst, int sw, int sh) {
  const int nw = (sw + 1) / 2, nh = (sh + 1) / 2;
  if (sw > 0 && sh > 0) hipLaunchKernelGGL(k_minify2, dim3((nw + 63) / 64, nh), dim3(64), 0, s, src, dst, sw, sh);
}
void launch_atlas_blit(hipStream_t s, uint32_t* level, int LS, int x, int y, const uint32_t* src, int w, int h) {
  if (w > 0 && h > 0) hipLaunchKernelGGL(k_atlas_blit, dim3((w + 63) / 64, h), dim3(64), 0, s, level, LS, x, y, src, w, h);
}

__global__ void k_fill_u32(uint32_t* p, uint32_t v, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = v;
}

void launch_fill(hipStream_t s, uint32_t* p, uint32_t v, size_t n) {
  if (n == 0) return;
  hipLaunchKernelGGL(k_fill_u32, dim3(1024), dim3(256), 0, s, p, v, n);
}

#if FDH_STATS
void debug_wave_times(unsigned long long* out) {
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wave_times), sizeof(unsigned long long) * 16 * 65536);
  void* p = nullptr;  // cleared after every read: the next read then holds exactly the launches in between
  if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_wave_times)) == hipSuccess) (void)hipMemset(p, 0, sizeof(unsigned long long) * 16 * 65536);
  (void)hipDeviceSynchronize();  // (the memset is asynchronous, the contexts' streams do not wait for the null stream: without this it clears rows of the NEXT launch)
}
void debug_counters(unsigned long long out[128], bool reset) {
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_counters), 128 * sizeof(unsigned long long));
  if (reset) { unsigned long long z[128] = {}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_counters), z, sizeof z); }
}
#endif

}  // namespace fdh
