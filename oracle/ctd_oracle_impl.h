/*
 * ctd_oracle_impl.h -- type-generic body of the CPU oracle (TEST INFRASTRUCTURE).
 *
 * Included twice by ctd_oracle.c with
 *     T      float / double
 *     SFX    f32   / f64
 *     SQRT   sqrtf / sqrt         (the reference compiles under torch/extension.h,
 *     FABS   fabsf / fabs          where unqualified sqrt/fabs pick the T overload)
 *
 * Every function restates one reference functor, with the SAME operation order,
 * the same int->T conversions and the same float/double promotions, so that a
 * build with -ffp-contract=off on x86-64 is bit-identical to the reference's
 * own ext_cpu.cpp (g++ -O3, no -mfma).  Citations are to /root/reference.
 */

#define CAT_(a, b) a##_##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SFX)

/* clamp helpers: mmax(0, mmin(n-1, v))  -- torchext/ext/common.h:68-86 */
static inline long FN(clampl)(long v, long n) {
  if (v > n - 1) v = n - 1;
  if (v < 0) v = 0;
  return v;
}

/* ------------------------------------------------------------------------- *
 * XCorrVolFunctor<T>::operator()   torchext/ext/ext.h:120-191
 * launch loop                      torchext/ext/ext_cpu.cpp:7-12, 88-105
 * in0,in1 [C,H,W]  ->  out [D,H,W]
 * ------------------------------------------------------------------------- */
int FN(ctd_oracle_xcorrvol)(const T* in0, const T* in1, T* out, long channels,
                            long height, long width, long n_disps,
                            long block_size, int nthreads) {
  const long N = n_disps * height * width;
  const long block_size2 = block_size * block_size;
  long oidx;
  (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
#endif
  for (oidx = 0; oidx < N; ++oidx) {
    long d = oidx / (height * width);          /* ext.h:136 */
    long h = (oidx / width) % height;          /* ext.h:137 */
    long w = oidx % width;                     /* ext.h:138 */
    T val = 0;
    for (int c = 0; c < channels; ++c) {
      T mu0 = 0, mu1 = 0;                      /* pass 1: means, ext.h:145-160 */
      for (int bh = 0; bh < block_size; ++bh) {
        long h0 = FN(clampl)(h + bh - block_size / 2, height);
        for (int bw = 0; bw < block_size; ++bw) {
          long w0 = w + bw - block_size / 2;
          long w1 = w0 - d;                    /* shifted BEFORE clamping, ext.h:152 */
          w0 = FN(clampl)(w0, width);
          w1 = FN(clampl)(w1, width);
          long idx0 = (c * height + h0) * width + w0;
          long idx1 = (c * height + h0) * width + w1;
          mu0 += in0[idx0] / (T)block_size2;   /* T / long -> T division, ext.h:157 */
          mu1 += in1[idx1] / (T)block_size2;
        }
      }
      T sigma0 = 0, sigma1 = 0, dot = 0;       /* pass 2, ext.h:163-183 */
      for (int bh = 0; bh < block_size; ++bh) {
        long h0 = FN(clampl)(h + bh - block_size / 2, height);
        for (int bw = 0; bw < block_size; ++bw) {
          long w0 = w + bw - block_size / 2;
          long w1 = w0 - d;
          w0 = FN(clampl)(w0, width);
          w1 = FN(clampl)(w1, width);
          long idx0 = (c * height + h0) * width + w0;
          long idx1 = (c * height + h0) * width + w1;
          T v0 = in0[idx0] - mu0;
          T v1 = in1[idx1] - mu1;
          dot += v0 * v1;
          sigma0 += v0 * v0;
          sigma1 += v1 * v1;
        }
      }
      /* sqrt in T, "+ 1e-8" in double, rounded back to T -- ext.h:185 */
      T norm = (T)((double)SQRT(sigma0 * sigma1) + 1e-8);
      val += dot / norm;                       /* ext.h:186 */
    }
    out[oidx] = val;
  }
  return 0;
}

/* argmax over the disparity axis of a [D,H,W] volume; first index wins ties
 * (== torch.argmax(vol, 0)); no reference code exists. */
int FN(ctd_oracle_argmax)(const T* vol, long long* idx, T* best, long n_disps,
                          long height, long width) {
  const long HW = height * width;
  for (long p = 0; p < HW; ++p) {
    T m = vol[p];
    long long mi = 0;
    for (long d = 1; d < n_disps; ++d) {
      T v = vol[d * HW + p];
      if (v > m) { m = v; mi = d; }
    }
    idx[p] = mi;
    if (best) best[p] = m;
  }
  return 0;
}

/* ------------------------------------------------------------------------- *
 * PhotometricLossForward<T,type>::operator()   torchext/ext/ext.h:201-266
 * host loop                                    ext_cpu.cpp:110-145
 * es,ta [B,C,H,W] -> out [B,1,H,W];  eps arrives as C float (ext_cpu.cpp:110)
 * ------------------------------------------------------------------------- */
int FN(ctd_oracle_photometric_fwd)(const T* es, const T* ta, T* out, int batch_size,
                                   int channels, int height, int width,
                                   int block_size, int type, float eps_f, int nthreads) {
  const int N = batch_size * height * width;
  const int block_size2 = block_size * block_size;
  const T eps = (T)eps_f;
  int outidx;
  if (type < 0 || type > 3) return 1;
  (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
#endif
  for (outidx = 0; outidx < N; ++outidx) {
    int w = outidx % width;
    int h = (outidx / width) % height;
    int n = outidx / (height * width);
    T loss = 0;
    for (int bidx = 0; bidx < block_size2; ++bidx) {
      int bh = bidx / block_size;
      int bw = bidx % block_size;
      int h0 = h + bh - block_size / 2;
      int w0 = w + bw - block_size / 2;
      h0 = h0 < 0 ? 0 : h0; h0 = h0 > height - 1 ? height - 1 : h0;   /* ext.h:231 */
      w0 = w0 < 0 ? 0 : w0; w0 = w0 > width - 1 ? width - 1 : w0;     /* ext.h:232 */
      for (int c = 0; c < channels; ++c) {
        int inidx = ((n * channels + c) * height + h0) * width + w0;
        if (type == 0 || type == 1) {
          T diff = es[inidx] - ta[inidx];
          if (type == 0) loss += diff * diff / (T)block_size2;        /* ext.h:239 */
          else           loss += FABS(diff) / (T)block_size2;         /* ext.h:242 */
        } else {
          int inidxc = ((n * channels + c) * height + h) * width + w;
          T des = es[inidx] - es[inidxc];
          T dta = ta[inidx] - ta[inidxc];
          /* "0.5 * (1 + x / sqrt(..))": inner part in T, the 0.5 multiply in
           * double, rounded to T (ext.h:249-250) */
          T h_des = (T)(0.5 * (double)((T)1 + des / SQRT(des * des + eps)));
          T h_dta = (T)(0.5 * (double)((T)1 + dta / SQRT(dta * dta + eps)));
          T diff = h_des - h_dta;
          if (type == 2) loss += diff * diff / (T)block_size2;        /* ext.h:255 */
          else           loss += FABS(diff) / (T)block_size2;         /* ext.h:258 */
        }
      }
    }
    out[outidx] = loss;
  }
  return 0;
}

/* ------------------------------------------------------------------------- *
 * PhotometricLossBackward<T,type>::operator()  torchext/ext/ext.h:268-344
 * host loop                                    ext_cpu.cpp:147-184
 * Serial scatter, outidx ascending, taps ascending: the accumulation order of
 * the reference CPU path.  grad_in is zero-filled here (ext_cpu.cpp:158).
 * ------------------------------------------------------------------------- */
int FN(ctd_oracle_photometric_bwd)(const T* es, const T* ta, const T* grad_out,
                                   T* grad_in, int batch_size, int channels,
                                   int height, int width, int block_size, int type,
                                   float eps_f) {
  const int N = batch_size * height * width;
  const int block_size2 = block_size * block_size;
  const T eps = (T)eps_f;
  if (type < 0 || type > 3) return 1;
  for (long i = 0; i < (long)batch_size * channels * height * width; ++i) grad_in[i] = 0;
  for (int outidx = 0; outidx < N; ++outidx) {
    int w = outidx % width;
    int h = (outidx / width) % height;
    int n = outidx / (height * width);
    for (int bidx = 0; bidx < block_size2; ++bidx) {
      int bh = bidx / block_size;
      int bw = bidx % block_size;
      int h0 = h + bh - block_size / 2;
      int w0 = w + bw - block_size / 2;
      h0 = h0 < 0 ? 0 : h0; h0 = h0 > height - 1 ? height - 1 : h0;
      w0 = w0 < 0 ? 0 : w0; w0 = w0 > width - 1 ? width - 1 : w0;
      const T go = grad_out[outidx];
      for (int c = 0; c < channels; ++c) {
        int inidx = ((n * channels + c) * height + h0) * width + w0;
        if (type == 0 || type == 1) {
          T diff = es[inidx] - ta[inidx];
          T grad = 0;
          if (type == 0) grad = (T)2 * diff;                               /* ext.h:309 */
          else grad = diff < 0 ? (T)-1 : (diff > 0 ? (T)1 : (T)0);         /* ext.h:312 */
          grad = grad / (T)block_size2 * go;                               /* ext.h:314 */
          grad_in[inidx] += grad;                                          /* ext.h:315 */
        } else {
          int inidxc = ((n * channels + c) * height + h) * width + w;
          T des = es[inidx] - es[inidxc];
          T dta = ta[inidx] - ta[inidxc];
          T h_des = (T)(0.5 * (double)((T)1 + des / SQRT(des * des + eps)));
          T h_dta = (T)(0.5 * (double)((T)1 + dta / SQRT(dta * dta + eps)));
          T diff = h_des - h_dta;
          T grad_loss = 0;
          if (type == 2) grad_loss = (T)2 * diff;                          /* ext.h:327 */
          else grad_loss = diff < 0 ? (T)-1 : (diff > 0 ? (T)1 : (T)0);    /* ext.h:330 */
          grad_loss = grad_loss / (T)block_size2;                          /* ext.h:332 */
          T tmp = des * des + eps;
          /* "0.5 * eps / sqrt(tmp^3)": sqrt in T, product and quotient in
           * double, rounded to T (ext.h:335) */
          T grad_heaviside = (T)(0.5 * (double)eps / (double)SQRT(tmp * tmp * tmp));
          T grad = go * grad_loss * grad_heaviside;                        /* ext.h:337 */
          grad_in[inidx] += grad;                                          /* ext.h:338 */
          grad_in[inidxc] += -grad;                                        /* ext.h:339 */
        }
      }
    }
  }
  return 0;
}


/* ------------------------------------------------------------------------- *
 * NNFunctor<T,3>::operator()   torchext/ext/ext.h:13-47   (host ext_cpu.cpp:14-36)
 * in0 [n0,3], in1 [n1,3] -> out int64 [n0]: index of the nearest in1 point
 * (squared distance accumulated d0,d1,d2 in order; strict <, so the first index
 * wins ties; -1 when nothing is closer than 1e9).
 * ------------------------------------------------------------------------- */
int FN(ctd_oracle_nn)(const T* in0, const T* in1, long nelem0, long nelem1, int64_t* out) {
  for (long idx0 = 0; idx0 < nelem0; ++idx0) {
    const T* vec0 = in0 + idx0 * 3;
    T min_dist = 1e9;                          /* ext.h:29 */
    long min_arg = -1;
    for (long idx1 = 0; idx1 < nelem1; ++idx1) {
      const T* vec1 = in1 + idx1 * 3;
      T dist = 0;
      for (long didx = 0; didx < 3; ++didx) {
        T diff = vec0[didx] - vec1[didx];
        dist += diff * diff;                   /* ext.h:36 */
      }
      if (dist < min_dist) {                   /* ext.h:39 */
        min_dist = dist;
        min_arg = idx1;
      }
    }
    out[idx0] = min_arg;
  }
  return 0;
}

/* ------------------------------------------------------------------------- *
 * ProjNNFunctor<T,3>::operator()   torchext/ext/ext.h:68-117   (host ext_cpu.cpp:59-86)
 * xyz0, xyz1 [B,H,W,3] (both in the coordinate system of view 1), K [3,3] ->
 * out int64 [B,H,W]: flat index (b*H + v)*W + u of the closest xyz1 point inside
 * the patch_size^2 patch around the projection of xyz0; -1 if the patch is empty.
 * `int u0 = u + 0.5` promotes to double and truncates toward zero (ext.h:94-95).
 * Projections that do not fit an int are undefined behaviour in the reference
 * (x86: INT_MIN, patch out of the image); restated as "no candidate".
 * ------------------------------------------------------------------------- */
int FN(ctd_oracle_proj_nn)(const T* xyz0, const T* xyz1, const T* K, long batch_size, long height,
                           long width, long patch_size, int64_t* out) {
  const long N = batch_size * height * width;
  for (long idx0 = 0; idx0 < N; ++idx0) {
    const long bs = idx0 / (height * width);
    const T x = xyz0[idx0 * 3 + 0];
    const T y = xyz0[idx0 * 3 + 1];
    const T z = xyz0[idx0 * 3 + 2];
    const T d = K[6] * x + K[7] * y + K[8] * z;
    const T u = (K[0] * x + K[1] * y + K[2] * z) / d;
    const T v = (K[3] * x + K[4] * y + K[5] * z) / d;
    const double ud = u + 0.5, vd = v + 0.5;
    long min_idx1 = -1;
    if (ud > -2147483649.0 && ud < 2147483648.0 && vd > -2147483649.0 && vd < 2147483648.0) {
      int u0 = (int)ud;
      int v0 = (int)vd;
      T min_dist = 1e9;
      for (int pidx = 0; pidx < patch_size * patch_size; ++pidx) {
        int pu = pidx % patch_size;
        int pv = pidx / patch_size;
        long u1 = (long)u0 + pu - patch_size / 2;
        long v1 = (long)v0 + pv - patch_size / 2;
        if (u1 >= 0 && v1 >= 0 && u1 < width && v1 < height) {
          const long idx1 = (bs * height + v1) * width + u1;
          const T* xyz1n = xyz1 + idx1 * 3;
          const T dd = (x - xyz1n[0]) * (x - xyz1n[0]) + (y - xyz1n[1]) * (y - xyz1n[1]) +
                       (z - xyz1n[2]) * (z - xyz1n[2]);          /* ext.h:108 */
          if (dd < min_dist) {
            min_dist = dd;
            min_idx1 = idx1;
          }
        }
      }
    }
    out[idx0] = min_idx1;
  }
  return 0;
}

#undef FN
#undef CAT
#undef CAT_
