// SURVEY.md 8(f) f2, image half: the dataset's RandomResizedCrop(448, bicubic) + RandomHorizontalFlip + Grayscale on the device
// (ECAMP/Pre-training/module/pretrain_datasets.py:47-52,113-115), bit for bit what the host transform produces.
//
// The reference runs torchvision 0.14.1's transforms on PIL images: F.resized_crop = img.crop(box).resize((448, 448), BICUBIC), i.e.
// Pillow's two-pass antialiased resample (Pillow 10.4.0 pinned in environment.yml:85; src/libImaging/Resample.c, unchanged in the
// 12.2.0 of this image).  Restated here from that published algorithm -- the filter support is widened by the scale factor, the
// normalised double-precision coefficients become 22-bit fixed point (precompute_coeffs / normalize_coeffs_8bpc), a horizontal pass
// writes a uint8 intermediate (ImagingResampleHorizontal_8bpc: 2^21 + sum(pixel * coef) >> 22, clipped), a vertical pass the same
// (ImagingResampleVertical_8bpc) -- and pinned by tests against Pillow itself (tests/test_augment*.py: equal bytes).
// MIMIC-CXR-JPG radiographs are grayscale: the reference's RGB crop has three equal channels, each resampled by the same integer
// arithmetic, and Grayscale's L = (19595 R + 38470 G + 7471 B + 2^15) >> 16 returns that common value -- so one uint8 plane is the
// whole item before ToTensor / Normalize (the `image_u8` schema the model's kernels already read).
//
// Host -> device: per sample only the bytes of the drawn crop box (contiguous h x w uint8) and a table row; the host draws the boxes
// and flips from the torch RNG in torchvision's order (ecamp_amd/module/pretrain_datasets.py: crop_params).
//   table[b] = {byte offset of the crop in `src`, h, w, flip, first row of the sample in the intermediate, unused}
// Three launches: coefficients (double precision, no contraction: the rounding of every operation is Pillow's), horizontal pass
// (rows x 448 uint8 intermediate in the caller's workspace), vertical pass (+ flip) -> dst uint8 [B, out, out].
#include "common.h"

#define RS_PRECISION_BITS 22

struct RsTab { long off, h, w, flip, row0, pad; };

__device__ __forceinline__ double rs_bicubic(double x) {
#pragma clang fp contract(off)
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}

// one thread per (sample, axis, output index): bounds {first input index, tap count} and `kmax` fixed-point taps (zero past the count)
__global__ __launch_bounds__(256) void resample_coef_kernel(const RsTab* __restrict__ tab, int2* __restrict__ bounds, int* __restrict__ coef, long B, int out,
                                                            int kmax, int* __restrict__ err) {
#pragma clang fp contract(off)
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= B * 2 * out) return;
    const int xx = (int)(id % out), axis = (int)((id / out) & 1);
    const long b = id / (2 * out);
    const int inSize = (int)(axis == 0 ? tab[b].w : tab[b].h);
    double scale = (double)inSize / (double)out, filterscale = scale;
    if (filterscale < 1.0) filterscale = 1.0;
    const double support = 2.0 * filterscale;
    const int ksize = (int)ceil(support) * 2 + 1;
    int* k = coef + id * kmax;
    if (ksize > kmax || inSize <= 0) {   // the host sized the tables for a smaller scale factor: refuse loudly (caller reads err)
        if (xx == 0) atomicExch(err, 1);
        bounds[id] = make_int2(0, 0);
        return;
    }
    const double center = 0 + (xx + 0.5) * scale;
    const double ss = 1.0 / filterscale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > inSize) xmax = inSize;
    xmax -= xmin;
    double ww = 0.0;
    for (int x = 0; x < xmax; ++x) ww += rs_bicubic((x + xmin - center + 0.5) * ss);
    for (int x = 0; x < kmax; ++x) {
        int q = 0;
        if (x < xmax) {
            double w = rs_bicubic((x + xmin - center + 0.5) * ss);
            if (ww != 0.0) w /= ww;
            q = w < 0 ? (int)(-0.5 + w * (double)(1 << RS_PRECISION_BITS)) : (int)(0.5 + w * (double)(1 << RS_PRECISION_BITS));
        }
        k[x] = q;
    }
    bounds[id] = make_int2(xmin, xmax);
}

__device__ __forceinline__ unsigned char rs_clip8(int v) {
    v >>= RS_PRECISION_BITS;
    return (unsigned char)(v < 0 ? 0 : v > 255 ? 255 : v);
}

// horizontal pass: block = (sample, chunk of ROWS rows); thread xx owns output column xx, its taps live in LDS as [tap][column]
#define RS_ROWS 32
__global__ __launch_bounds__(512) void resample_h_kernel(const unsigned char* __restrict__ src, const RsTab* __restrict__ tab, const int2* __restrict__ bounds,
                                                         const int* __restrict__ coef, unsigned char* __restrict__ tmp, int out, int kmax) {
    extern __shared__ int kl[];   // [kmax][out]
    const long b = blockIdx.y;
    const RsTab t = tab[b];
    const int r0 = blockIdx.x * RS_ROWS;
    if (r0 >= t.h) return;
    const long cid = (b * 2 + 0) * out;
    for (int i = threadIdx.x; i < out * kmax; i += blockDim.x) {
        const int xx = i / kmax, tp = i - xx * kmax;
        kl[tp * out + xx] = coef[(cid + xx) * kmax + tp];
    }
    __syncthreads();
    const int xx = threadIdx.x;
    if (xx >= out) return;
    const int2 bd = bounds[cid + xx];
    const int r1 = min(r0 + RS_ROWS, (int)t.h);
    for (int r = r0; r < r1; ++r) {
        const unsigned char* p = src + t.off + (long)r * t.w + bd.x;
        int ss = 1 << (RS_PRECISION_BITS - 1);
        for (int tp = 0; tp < bd.y; ++tp) ss += (int)p[tp] * kl[tp * out + xx];
        tmp[(t.row0 + r) * out + xx] = rs_clip8(ss);
    }
}

// vertical pass (+ horizontal flip): block = (sample, output row); the row's taps are the same for every thread
__global__ __launch_bounds__(512) void resample_v_kernel(const unsigned char* __restrict__ tmp, const RsTab* __restrict__ tab, const int2* __restrict__ bounds,
                                                         const int* __restrict__ coef, unsigned char* __restrict__ dst, int out, int kmax) {
    const long b = blockIdx.y;
    const int yy = blockIdx.x, xx = threadIdx.x;
    if (xx >= out) return;
    const RsTab t = tab[b];
    const long cid = (b * 2 + 1) * out + yy;
    const int2 bd = bounds[cid];
    const int* k = coef + cid * kmax;
    const unsigned char* p = tmp + (t.row0 + bd.x) * out + xx;
    int ss = 1 << (RS_PRECISION_BITS - 1);
    for (int tp = 0; tp < bd.y; ++tp) ss += (int)p[(long)tp * out] * k[tp];
    dst[(b * out + yy) * out + (t.flip ? out - 1 - xx : xx)] = rs_clip8(ss);
}

// workspace: bounds int2 [B][2][out] | coef int32 [B][2][out][kmax] | err int32 (16 B) | intermediate uint8 [tmp_rows][out]
static size_t rs_align(size_t v) { return (v + 255) & ~(size_t)255; }
extern "C" int64_t ecamp_resample_crops_workspace_bytes(int64_t B, int32_t out, int32_t kmax, int64_t tmp_rows) {
    if (B <= 0 || out <= 0 || kmax <= 0 || tmp_rows <= 0) return 0;
    return (int64_t)(rs_align((size_t)B * 2 * out * sizeof(int2)) + rs_align((size_t)B * 2 * out * kmax * sizeof(int)) + 256 + rs_align((size_t)tmp_rows * out));
}

extern "C" int ecamp_resample_crops_u8(const uint8_t* src, const int64_t* table, uint8_t* dst, int64_t B, int32_t out, int32_t kmax, int64_t tmp_rows,
                                       int32_t max_h, void* ws, int64_t ws_bytes, int32_t* err_flag, hipStream_t stream) {
    ECAMP_CHECK_ARG(src && table && dst && ws && err_flag, "resample_crops_u8: null argument");
    ECAMP_CHECK_ARG(B > 0 && out > 0 && out <= 512 && kmax >= 5 && (size_t)kmax * out * sizeof(int) <= 160 * 1024 && tmp_rows > 0 && max_h > 0 && max_h <= tmp_rows,
                    "resample_crops_u8: B=%ld out=%d (<= 512) kmax=%d (taps x out x 4 B must fit the 160 KB LDS) tmp_rows=%ld max_h=%d", (long)B, out, kmax, (long)tmp_rows, max_h);
    ECAMP_CHECK_ARG(ws_bytes >= ecamp_resample_crops_workspace_bytes(B, out, kmax, tmp_rows), "resample_crops_u8: workspace of %ld bytes is too small", (long)ws_bytes);
    unsigned char* w = reinterpret_cast<unsigned char*>(ws);
    int2* bounds = reinterpret_cast<int2*>(w);
    w += rs_align((size_t)B * 2 * out * sizeof(int2));
    int* coef = reinterpret_cast<int*>(w);
    w += rs_align((size_t)B * 2 * out * kmax * sizeof(int));
    (void)w;   // (the 256 bytes behind the tables are spare)
    unsigned char* tmp = w + 256;
    const RsTab* tab = reinterpret_cast<const RsTab*>(table);
    const long n = B * 2 * out;
    hipLaunchKernelGGL(resample_coef_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, tab, bounds, coef, (long)B, (int)out, (int)kmax, (int*)err_flag);
    // rows per sample are data (the table lives on the device): the grid covers the tallest crop of the batch (max_h, the host knows it)
    const size_t shm = (size_t)kmax * out * sizeof(int);
    static bool optin = false;
    if (!optin) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(resample_h_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); optin = true; }
    const unsigned chunks = (unsigned)((max_h + RS_ROWS - 1) / RS_ROWS);
    hipLaunchKernelGGL(resample_h_kernel, dim3(chunks, (unsigned)B), dim3(512), shm, stream, src, tab, bounds, coef, tmp, (int)out, (int)kmax);
    hipLaunchKernelGGL(resample_v_kernel, dim3((unsigned)out, (unsigned)B), dim3(512), 0, stream, tmp, tab, bounds, coef, dst, (int)out, (int)kmax);
    ECAMP_LAUNCH_CHECK();
    return 0;
}
