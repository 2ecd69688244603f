// Scene compositing around the network: cutout gather and the residual / predicted fields.
//
// Reference behaviour restated:
//  * extract/extraction.py:4-43       cutout i = field[0, xs:xs+cs, ys:ys+cs, :] (callers validate the window)
//  * deblend/field_deblender.py:46-97   residual  = field - sum_i shift(pad(mean_i), (x_i, y_i))        (in object order)
//  * deblend/field_deblender.py:99-189  predicted = sum_i shift(pad(stamp_i), (x_i, y_i)) for mean / stddev / epistemic
//    where pad() centres the cs x cs stamp in a zero F x F image and shift() is scipy.ndimage.shift with its defaults
//    (cubic B-spline, mode "constant", prefilter on).  The reference builds an F x F image per object and band and
//    shifts it on the CPU (its own TODO at :82 calls this "super slow").
//
// Here: float64 like the reference's numpy arrays; one thread per field element walks the objects IN ORDER (same
// summation order as the reference loop, so results are deterministic and agree to rounding).  An object whose two
// shifts are integers is an exact translation (no spline needed: interpolating a spline at its knots returns the
// samples).  Other objects get cubic B-spline coefficients of the zero-extended stamp (recursive prefilter, pole
// sqrt(3)-2, margin T = 20 pixels: 0.268^20 = 4e-12) and are evaluated with the 4 x 4 B-spline weights.
// HBM-bound byte work; nothing here wants MFMA.
#include "common.h"

namespace dv {

namespace {
constexpr int T_MARGIN = 20;

struct SceneObj {
  double sx, sy;     // shift of the stamp's top-left corner relative to field index 0 (po + pos), rows / cols
  int ix, iy;        // the same as integers (integer objects)
  int coef;          // index into the coefficient buffer, -1 for integer objects
  int pad_;
};

// Cutout gather (extract/extraction.py:4-43), one workgroup per cutout; a wave copies every fourth row, 64 consecutive
// elements (pixel-major, band-minor) per trip: coalesced on both sides and no index arithmetic beyond one multiply per row
// (the first version decoded a flat 64-bit element index with four divisions per element: 1.08 ms per 8192-cutout chunk of
// the inference pipeline, 0.41 ms now).  OUT = double: the reference's cutouts; OUT = float: cast as deblend() does
// (tf.cast, deblender.py:18), straight into the network's input buffer.
template <typename OUT>
__global__ __launch_bounds__(256) void scene_extract_kernel(const double* __restrict__ field, int F, int nb,
                                                            const int* __restrict__ starts, int cs, OUT* __restrict__ out) {
  const int n = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int x0 = starts[2 * n], y0 = starts[2 * n + 1];
  const int rowlen = cs * nb;
  OUT* o = out + (long)n * cs * rowlen;
  for (int i = wave; i < cs; i += 4) {
    const double* src = field + ((long)(x0 + i) * F + y0) * nb;
    OUT* dst = o + (long)i * rowlen;
    for (int x = lane; x < rowlen; x += 64) dst[x] = (OUT)src[x];
  }
}

// cubic B-spline coefficients of stamp `objs[o]` zero-extended by T on every side: coef[k][P][P][nb]
__global__ __launch_bounds__(256) void scene_prefilter_kernel(const double* __restrict__ stamps,
                                                              const int* __restrict__ which, int cs, int nb,
                                                              double* __restrict__ coef) {
  const int P = cs + 2 * T_MARGIN;
  const double z1 = -0.26794919243112270647;   // sqrt(3) - 2
  const double* st = stamps + (long)which[blockIdx.x] * cs * cs * nb;
  double* c = coef + (long)blockIdx.x * P * P * nb;
  // axis 0 (rows) for the cs stamp columns; the other columns of the extended image are zero and stay zero
  for (int t = threadIdx.x; t < cs * nb; t += 256) {
    const int j = t / nb, b = t - j * nb;
    double* col = c + ((long)(T_MARGIN + j)) * nb + b;          // element [k][T+j][b] at col[k * P * nb]
    const long ks = (long)P * nb;
    double acc = 0.0;
    for (int k = 0; k < P; ++k) {
      const int i = k - T_MARGIN;
      const double s = (i >= 0 && i < cs) ? st[((long)i * cs + j) * nb + b] : 0.0;
      acc = 6.0 * s + z1 * acc;
      col[k * ks] = acc;
    }
    double nxt = 0.0;
    for (int k = P - 1; k >= 0; --k) {
      nxt = z1 * (nxt - col[k * ks]);
      col[k * ks] = nxt;
    }
  }
  __syncthreads();
  // axis 1 (columns) for every row
  for (int t = threadIdx.x; t < P * nb; t += 256) {
    const int k = t / nb, b = t - k * nb;
    double* row = c + (long)k * P * nb + b;                       // element [k][j][b] at row[j * nb]
    double acc = 0.0;
    for (int j = 0; j < P; ++j) {
      const int jj = j - T_MARGIN;
      const double s = (jj >= 0 && jj < cs) ? row[(long)j * nb] : 0.0;
      acc = 6.0 * s + z1 * acc;
      row[(long)j * nb] = acc;
    }
    double nxt = 0.0;
    for (int j = P - 1; j >= 0; --j) {
      nxt = z1 * (nxt - row[(long)j * nb]);
      row[(long)j * nb] = nxt;
    }
  }
}

__device__ __forceinline__ void bspline3(double t, double w[4]) {
  const double u = 1.0 - t;
  w[0] = u * u * u / 6.0;
  w[1] = (3.0 * t * t * t - 6.0 * t * t + 4.0) / 6.0;
  w[2] = (-3.0 * t * t * t + 3.0 * t * t + 3.0 * t + 1.0) / 6.0;
  w[3] = t * t * t / 6.0;
}

__global__ __launch_bounds__(256) void scene_composite_kernel(double* __restrict__ field, int F, int nb,
                                                              const double* __restrict__ stamps,
                                                              const double* __restrict__ coef,
                                                              const SceneObj* __restrict__ objs, int nobj, int cs,
                                                              int po, double sign) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long)F * F * nb) return;
  const int b = (int)(e % nb);
  const long px = e / nb;
  const int c = (int)(px % F), r = (int)(px / F);
  const int P = cs + 2 * T_MARGIN;
  double acc = field[e];
  for (int o = 0; o < nobj; ++o) {
    const SceneObj ob = objs[o];
    if (ob.coef < 0) {
      const int rr = r - ob.ix, cc = c - ob.iy;
      if ((unsigned)rr < (unsigned)cs && (unsigned)cc < (unsigned)cs)
        acc += sign * stamps[(((long)o * cs + rr) * cs + cc) * nb + b];
      continue;
    }
    // scipy.ndimage.shift, mode "constant": output is cval where the input coordinate leaves [0, F-1]; the spline
    // coefficients are those of the F x F padded image with MIRROR boundaries at its edge samples, and nodes
    // outside the image are looked up mirrored.  With the infinite-domain coefficients cinf of the zero-extended
    // stamp (what the prefilter kernel computes) the mirrored ones are cm[i] = cinf[i] + cinf[-i] + cinf[2(F-1)-i].
    // The reflected terms vanish unless the padded stamp sits within ~T pixels of the image edge (po < T).
    const double xin = (double)r - (ob.sx - po), yin = (double)c - (ob.sy - po);
    if (xin < 0.0 || yin < 0.0 || xin > F - 1.0 || yin > F - 1.0) continue;
    const int off = po - T_MARGIN;                 // padded-image index of coefficient-image index 0
    const bool refl = off < 2;
    if (!refl && (xin - off < -2.0 || yin - off < -2.0 || xin - off > P + 1.0 || yin - off > P + 1.0)) continue;
    const double fx = floor(xin), fy = floor(yin);
    double wx[4], wy[4];
    bspline3(xin - fx, wx);
    bspline3(yin - fy, wy);
    const double* cf = coef + (long)ob.coef * P * P * nb + b;
    const int nr = refl ? 3 : 1;
    double v = 0.0;
    for (int a = 0; a < 4; ++a) {
      int i = (int)fx - 1 + a;
      i = i < 0 ? -i : (i > F - 1 ? 2 * (F - 1) - i : i);
      const int ri[3] = {i - off, -i - off, 2 * (F - 1) - i - off};
      for (int d = 0; d < 4; ++d) {
        int j = (int)fy - 1 + d;
        j = j < 0 ? -j : (j > F - 1 ? 2 * (F - 1) - j : j);
        const int rj[3] = {j - off, -j - off, 2 * (F - 1) - j - off};
        double cm = 0.0;
        for (int u = 0; u < nr; ++u) {
          if ((unsigned)ri[u] >= (unsigned)P) continue;
          for (int w = 0; w < nr; ++w)
            if ((unsigned)rj[w] < (unsigned)P) cm += cf[((long)ri[u] * P + rj[w]) * nb];
        }
        v += wx[a] * wy[d] * cm;
      }
    }
    acc += sign * v;
  }
  field[e] = acc;
}

// ---- compositing of one inference chunk, device resident (dv_infer_cutouts_composite) --------------------------------
// field += stamp_i placed with its top-left corner at places[i] = (row, col), for the n stamps of a chunk IN OBJECT ORDER
// per field element - the order of the reference's loop over res_deblend (field_deblender.py:112-183: one
// scipy.ndimage.shift of a padded image per object, integer shifts) and of scene_composite_kernel above, so the sums are
// bit-identical to the host-composited path.  Stamps are the network's float32 outputs as the forward pass left them in
// HBM (mean and stddev of every stamp, 167 KB per stamp that never cross the host link).
//
// A workgroup owns a 32 x 32-pixel tile of the field (four pixels per thread) and scans the chunk's objects in rounds of 2048 (eight per thread,
// their placements by four 16-byte loads): the objects whose window meets the tile are compacted IN ORDER into an LDS
// list (a prefix sum of the hit counts over the 256 threads), then every thread walks the list for its pixel.  Uniformly
// scattered cutouts leave ~4 entries per round and tile; a pile of objects on one spot just makes the lists long (a
// round's list holds all 2048) - no capacity limit, no atomics, no float non-determinism.
constexpr int CT = 32;       // tile edge: a thread owns the four pixels (ty + 16 a, tx + 16 b) of its 32 x 32 tile
constexpr int CSEG = 2048;   // objects per scan round (8 per thread)
template <int NBMAX>
__global__ __launch_bounds__(256) void scene_composite_chunk_kernel(double* __restrict__ mean_f, double* __restrict__ std_f,
                                                                    double* __restrict__ res_f, int F, int nb,
                                                                    const float* __restrict__ loc,
                                                                    const float* __restrict__ scale,
                                                                    const int* __restrict__ places, int n, int cs) {
  __shared__ int s_list[CSEG];        // objects of the round that meet the tile, in object order
  __shared__ int s_lr[CSEG], s_lc[CSEG];   // their placements
  __shared__ int s_wsum[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ntx = (F + CT - 1) / CT;
  const int tr0 = (blockIdx.x / ntx) * CT, tc0 = (blockIdx.x % ntx) * CT;
  const int ty = tid >> 4, tx = tid & 15;
  double am[4][NBMAX], as[4][NBMAX], ar[4][NBMAX];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int b = 0; b < NBMAX; ++b) am[q][b] = as[q][b] = ar[q][b] = 0.0;
  bool loaded = false;
  unsigned touched = 0;
  for (int seg = 0; seg < n; seg += CSEG) {
    // thread t tests objects seg + 8 t .. + 7 (four 16-byte loads of their placements): order by (thread, bit) = object order
    const int o0 = seg + tid * 8;
    unsigned hits = 0;
    int pr[8], pc[8];
    if (o0 + 8 <= n && (reinterpret_cast<size_t>(places + 2 * o0) & 15) == 0) {
      const int4* q = reinterpret_cast<const int4*>(places + 2 * o0);
      const int4 a = q[0], b = q[1], d = q[2], e = q[3];
      pr[0] = a.x; pc[0] = a.y; pr[1] = a.z; pc[1] = a.w; pr[2] = b.x; pc[2] = b.y; pr[3] = b.z; pc[3] = b.w;
      pr[4] = d.x; pc[4] = d.y; pr[5] = d.z; pc[5] = d.w; pr[6] = e.x; pc[6] = e.y; pr[7] = e.z; pc[7] = e.w;
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const bool v = o0 + k < n;
        pr[k] = v ? places[2 * (o0 + k)] : (1 << 29);
        pc[k] = v ? places[2 * (o0 + k) + 1] : (1 << 29);
      }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (pr[k] < tr0 + CT && pr[k] + cs > tr0 && pc[k] < tc0 + CT && pc[k] + cs > tc0) hits |= 1u << k;
    // exclusive prefix of the hit counts over the 256 threads: wave scan, then the wave totals through LDS
    const int cnt = __popc(hits);
    int incl = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int v = __shfl_up(incl, d, 64);
      if (lane >= d) incl += v;
    }
    if (lane == 63) s_wsum[wave] = incl;
    __syncthreads();
    int base = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w)
      if (w < wave) base += s_wsum[w];
    const int total = s_wsum[0] + s_wsum[1] + s_wsum[2] + s_wsum[3];
    int pos = base + incl - cnt;
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (hits & (1u << k)) {
        s_list[pos] = o0 + k;
        s_lr[pos] = pr[k];
        s_lc[pos] = pc[k];
        ++pos;
      }
    __syncthreads();
    if (total > 0) {
      if (!loaded) {
        loaded = true;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int r = tr0 + ty + 16 * (q >> 1), c = tc0 + tx + 16 * (q & 1);
          if (r < F && c < F) {
            const long e0 = ((long)r * F + c) * nb;
#pragma unroll
            for (int b = 0; b < NBMAX; ++b)
              if (b < nb) {
                am[q][b] = mean_f[e0 + b];
                as[q][b] = std_f[e0 + b];
                if (res_f) ar[q][b] = res_f[e0 + b];
              }
          }
        }
      }
      for (int k = 0; k < total; ++k) {
        const int lr = s_lr[k], lc = s_lc[k];
        const long sb = (long)s_list[k] * cs;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int rr = tr0 + ty + 16 * (q >> 1) - lr, cc = tc0 + tx + 16 * (q & 1) - lc;
          if ((unsigned)rr < (unsigned)cs && (unsigned)cc < (unsigned)cs) {
            const long so = ((sb + rr) * cs + cc) * nb;
            touched |= 1u << q;
#pragma unroll
            for (int b = 0; b < NBMAX; ++b)
              if (b < nb) {
                const double v = (double)loc[so + b];
                am[q][b] += v;
                ar[q][b] -= v;
                as[q][b] += (double)scale[so + b];
              }
          }
        }
      }
    }
    __syncthreads();                       // the list is rewritten by the next round
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int r = tr0 + ty + 16 * (q >> 1), c = tc0 + tx + 16 * (q & 1);
    if ((touched >> q) & 1u) {             // (touched implies the pixel lies inside the field: windows are tested against it)
      if (r < F && c < F) {
        const long e0 = ((long)r * F + c) * nb;
#pragma unroll
        for (int b = 0; b < NBMAX; ++b)
          if (b < nb) {
            mean_f[e0 + b] = am[q][b];
            std_f[e0 + b] = as[q][b];
            if (res_f) res_f[e0 + b] = ar[q][b];
          }
      }
    }
  }
}

// mse_center[i] = mean over the centre 10 x 10 pixels and all bands of (cutout_i - mean_i)^2 in float64
// (field_deblender.py:323-327: mse(cutout_images[k, c0:c1, c0:c1], output_images_mean[i, c0:c1, c0:c1]) with
// c0 = int(cs/2) - 5, c1 = int(cs/2) + 5; training/metrics.py:4-12); one wave per stamp
__global__ __launch_bounds__(256) void scene_center_mse_kernel(const double* __restrict__ field, int F, int nb,
                                                               const int* __restrict__ starts,
                                                               const float* __restrict__ loc, int n, int cs,
                                                               double* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  const int c0 = cs / 2 - 5, w = 10;
  const int x0 = starts[2 * i], y0 = starts[2 * i + 1];
  double acc = 0.0;
  const int total = w * w * nb;
  for (int e = lane; e < total; e += 64) {
    const int b = e % nb, q = e / nb, cc = q % w, rr = q / w;
    const double a = field[((long)(x0 + c0 + rr) * F + (y0 + c0 + cc)) * nb + b];
    const double m = (double)loc[(((long)i * cs + c0 + rr) * cs + c0 + cc) * nb + b];
    acc += (a - m) * (a - m);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if (lane == 0) out[i] = acc / (double)total;
}
}  // namespace

int scene_extract(const double* field_h, int F, int nb, const int32_t* starts_h, int N, int cs, double* out_h,
                  hipStream_t s) {
  if (!field_h || !starts_h || !out_h || F < 1 || nb < 1 || cs < 1 || N < 0) {
    set_error("scene_extract: bad arguments");
    return E_INVALID;
  }
  if (N == 0) return OK;
  for (int i = 0; i < N; ++i) {
    const int x = starts_h[2 * i], y = starts_h[2 * i + 1];
    if (x < 0 || y < 0 || x > F - cs || y > F - cs) {   // (cs <= F holds; no x + cs: it overflows near INT_MAX)
      set_error("scene_extract: cutout %d (start %d,%d size %d) leaves the %d-pixel field", i, x, y, cs, F);
      return E_INVALID;
    }
  }
  const size_t fb = (size_t)F * F * nb * sizeof(double), ob = (size_t)N * cs * cs * nb * sizeof(double);
  double *field = nullptr, *out = nullptr;
  int* starts = nullptr;
  int st = OK;
  auto cleanup = [&]() { (void)hipFree(field); (void)hipFree(out); (void)hipFree(starts); };
#define SC_HIP(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) { st = hip_fail(e__, #call, __FILE__, __LINE__); cleanup(); return st; } } while (0)
  SC_HIP(hipMalloc((void**)&field, fb));
  SC_HIP(hipMalloc((void**)&out, ob));
  SC_HIP(hipMalloc((void**)&starts, (size_t)N * 2 * sizeof(int)));
  SC_HIP(hipMemcpyAsync(field, field_h, fb, hipMemcpyHostToDevice, s));
  SC_HIP(hipMemcpyAsync(starts, starts_h, (size_t)N * 2 * sizeof(int), hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(scene_extract_kernel<double>, dim3((unsigned)N), dim3(256), 0, s, field, F, nb, starts, cs, out);
  SC_HIP(hipGetLastError());
  SC_HIP(hipMemcpyAsync(out_h, out, ob, hipMemcpyDeviceToHost, s));
  SC_HIP(hipStreamSynchronize(s));
  cleanup();
  return OK;
}

int launch_scene_extract_f32(const double* field_dev, int F, int nb, const int* starts_dev, long count, int cs,
                             float* out_dev, hipStream_t s) {
  const long total = count * cs * cs * nb;
  if (total <= 0) return OK;
  hipLaunchKernelGGL(scene_extract_kernel<float>, dim3((unsigned)count), dim3(256), 0, s, field_dev, F, nb, starts_dev, cs,
                     out_dev);
  DV_HIP(hipGetLastError());
  return OK;
}

int scene_composite(double* field_h, int F, int nb, const double* stamps_h, const double* pos_h, int N, int cs,
                    double sign, hipStream_t s) {
  if (!field_h || F < 1 || nb < 1 || cs < 1 || cs > F || N < 0 || (N > 0 && (!stamps_h || !pos_h))) {
    set_error("scene_composite: bad arguments");
    return E_INVALID;
  }
  if (N == 0) return OK;
  const int P = cs + 2 * T_MARGIN;
  const int po = (F - cs) / 2;                 // int((field_size - cutout_size) / 2), field_deblender.py:70
  const int CHUNK = 256;                       // objects per pass (coefficient buffer <= 256 * P*P*nb doubles)
  const size_t fb = (size_t)F * F * nb * sizeof(double);
  const size_t stamp_elems = (size_t)cs * cs * nb;
  double *field = nullptr, *stamps = nullptr, *coef = nullptr;
  SceneObj* objs = nullptr;
  int* which = nullptr;
  int st = OK;
  auto cleanup = [&]() {
    (void)hipFree(field); (void)hipFree(stamps); (void)hipFree(coef); (void)hipFree(objs); (void)hipFree(which);
  };
  SC_HIP(hipMalloc((void**)&field, fb));
  SC_HIP(hipMalloc((void**)&stamps, (size_t)CHUNK * stamp_elems * sizeof(double)));
  SC_HIP(hipMalloc((void**)&objs, (size_t)CHUNK * sizeof(SceneObj)));
  SC_HIP(hipMalloc((void**)&which, (size_t)CHUNK * sizeof(int)));
  SC_HIP(hipMemcpyAsync(field, field_h, fb, hipMemcpyHostToDevice, s));
  SceneObj hobj[256];
  int hwhich[256];
  for (int base = 0; base < N; base += CHUNK) {
    const int n = N - base < CHUNK ? N - base : CHUNK;
    int nsub = 0;
    for (int i = 0; i < n; ++i) {
      const double px = pos_h[2 * (base + i)], py = pos_h[2 * (base + i) + 1];
      if (!(px == px) || !(py == py) || px > 1e9 || px < -1e9 || py > 1e9 || py < -1e9) {
        set_error("scene_composite: object %d has a non-finite position", base + i);
        cleanup();
        return E_INVALID;
      }
      SceneObj o;
      o.sx = po + px; o.sy = po + py;
      const bool integer = px == floor(px) && py == floor(py);
      o.ix = (int)floor(o.sx); o.iy = (int)floor(o.sy);
      o.coef = integer ? -1 : nsub;
      o.pad_ = 0;
      if (!integer) hwhich[nsub++] = i;
      hobj[i] = o;
    }
    SC_HIP(hipMemcpyAsync(stamps, stamps_h + (size_t)base * stamp_elems, (size_t)n * stamp_elems * sizeof(double),
                          hipMemcpyHostToDevice, s));
    SC_HIP(hipMemcpyAsync(objs, hobj, (size_t)n * sizeof(SceneObj), hipMemcpyHostToDevice, s));
    if (nsub > 0) {
      if (!coef) SC_HIP(hipMalloc((void**)&coef, (size_t)CHUNK * P * P * nb * sizeof(double)));
      SC_HIP(hipMemcpyAsync(which, hwhich, (size_t)nsub * sizeof(int), hipMemcpyHostToDevice, s));
      hipLaunchKernelGGL(scene_prefilter_kernel, dim3(nsub), dim3(256), 0, s, stamps, which, cs, nb, coef);
      SC_HIP(hipGetLastError());
    }
    const long total = (long)F * F * nb;
    hipLaunchKernelGGL(scene_composite_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, field, F, nb,
                       stamps, coef, objs, n, cs, po, sign);
    SC_HIP(hipGetLastError());
    SC_HIP(hipStreamSynchronize(s));           // hobj / hwhich are reused by the next chunk
  }
  SC_HIP(hipMemcpyAsync(field_h, field, fb, hipMemcpyDeviceToHost, s));
  SC_HIP(hipStreamSynchronize(s));
  cleanup();
  return OK;
#undef SC_HIP
}

}  // namespace dv

namespace dv {
int launch_scene_composite_chunk(double* mean_f, double* std_f, double* res_f, int F, int nb, const float* loc,
                                 const float* scale, const int* places_dev, int n, int cs, hipStream_t s) {
  if (n <= 0) return OK;
  if (nb < 1 || nb > 8) {
    set_error("scene composite: 1 .. 8 bands");
    return E_INVALID;
  }
  const int ntx = (F + CT - 1) / CT;
  if (nb <= 6)
    hipLaunchKernelGGL(scene_composite_chunk_kernel<6>, dim3((unsigned)(ntx * ntx)), dim3(256), 0, s, mean_f, std_f, res_f, F,
                       nb, loc, scale, places_dev, n, cs);
  else
    hipLaunchKernelGGL(scene_composite_chunk_kernel<8>, dim3((unsigned)(ntx * ntx)), dim3(256), 0, s, mean_f, std_f, res_f, F,
                       nb, loc, scale, places_dev, n, cs);
  DV_HIP(hipGetLastError());
  return OK;
}

int launch_scene_center_mse(const double* field_dev, int F, int nb, const int* starts_dev, const float* loc, int n, int cs,
                            double* out_dev, hipStream_t s) {
  if (n <= 0) return OK;
  if (cs < 10) {
    set_error("centre MSE needs stamps of at least 10 pixels");
    return E_INVALID;
  }
  hipLaunchKernelGGL(scene_center_mse_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, field_dev, F, nb, starts_dev, loc,
                     n, cs, out_dev);
  DV_HIP(hipGetLastError());
  return OK;
}
}  // namespace dv
