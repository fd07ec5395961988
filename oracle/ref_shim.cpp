/*
 * oracle/ref_shim.cpp -- host-pointer C entry points around the REFERENCE rasterizer itself.
 *
 * TEST INFRASTRUCTURE ONLY (tests/ and tools/ may load the library this builds; the product never does).
 *
 * oracle/build_ref.sh runs hipify-perl over the reference's own sources where they lie
 * (/root/reference/submodules/depth-diff-gaussian-rasterization/cuda_rasterizer/{forward,backward,rasterizer_impl}.cu
 * + headers) into a temporary directory, compiles them for gfx950 and links them with this file into
 * oracle/_ref/libref_raster*.so (git-ignored; ships to the GPU box like libs3g.so).  No reference source enters the
 * repository.  This file replaces RAST/rasterize_points.cu:35-202 (the torch-tensor wrapper) by plain host pointers:
 * it uploads the inputs, calls CudaRasterizer::Rasterizer::{forward,backward,markVisible} (RAST/cuda_rasterizer/
 * rasterizer.h:20-88) with hipMalloc-backed resize callbacks, and downloads outputs plus the internal state
 * (GeometryState / ImageState / BinningState, RAST/cuda_rasterizer/rasterizer_impl.h:29-63) that the parity tests
 * compare: radii, depths, means2D, cov3D, conic_opacity, rgb, clamped, tiles_touched, point_offsets, ranges,
 * n_contrib, final_T (accum_alpha), sorted keys and point_list.
 *
 * Struct layouts equal those of oracle/raster_oracle.c (REAL=float) so oracle/oracle.py's ctypes structs are shared.
 */
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#include "rasterizer_impl.h" /* hipified reference header, from the temporary build directory */

namespace {

struct Inputs {
    int P, D, M, W, H;
    const float *background, *means3D, *shs, *colors_precomp, *opacities, *scales;
    float scale_modifier;
    const float *rotations, *cov3D_precomp, *viewmatrix, *projmatrix, *cam_pos;
    float tan_fovx, tan_fovy;
};

struct State {
    float* depths;
    uint8_t* clamped;
    float *means2D, *cov3D, *conic_opacity, *rgb;
    uint32_t *tiles_touched, *point_offsets;
    float* final_T;
    uint32_t *n_contrib, *ranges;
    uint64_t* point_list_keys;
    uint32_t* point_list;
    int num_rendered;
};

#define HIPCHK(x)                                                                              \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) {                                                                \
            fprintf(stderr, "ref_shim: %s -> %s\n", #x, hipGetErrorString(e_));               \
            abort();                                                                           \
        }                                                                                      \
    } while (0)

struct Arena {
    char* ptr = nullptr;
    size_t cap = 0;
    std::function<char*(size_t)> fn() {
        return [this](size_t n) {
            if (n > cap) {
                if (ptr) HIPCHK(hipFree(ptr));
                HIPCHK(hipMalloc(&ptr, n));
                cap = n;
            }
            return ptr;
        };
    }
    ~Arena() {
        if (ptr) (void)hipFree(ptr);
    }
};

struct Handle {
    Inputs cfg;
    int R = 0;
    Arena geom, binning, img;
    std::vector<void*> dev;
    float *bg = nullptr, *means3D = nullptr, *shs = nullptr, *colors = nullptr, *opac = nullptr, *scales = nullptr,
          *rots = nullptr, *cov = nullptr, *view = nullptr, *proj = nullptr, *campos = nullptr;
    int* radii = nullptr;
    float* up(const float* h, size_t n) {
        if (!h || n == 0) return nullptr;
        float* d;
        HIPCHK(hipMalloc(&d, n * sizeof(float)));
        HIPCHK(hipMemcpy(d, h, n * sizeof(float), hipMemcpyHostToDevice));
        dev.push_back(d);
        return d;
    }
    template <typename T>
    T* zeros(size_t n) {
        T* d;
        HIPCHK(hipMalloc(&d, (n ? n : 1) * sizeof(T)));
        HIPCHK(hipMemset(d, 0, (n ? n : 1) * sizeof(T)));
        dev.push_back(d);
        return d;
    }
    ~Handle() {
        for (void* p : dev) (void)hipFree(p);
    }
};

template <typename T>
void down(T* host, const T* devp, size_t n) {
    if (host && n) HIPCHK(hipMemcpy(host, devp, n * sizeof(T), hipMemcpyDeviceToHost));
}

}  // namespace

extern "C" {

int ref_sizeof_real(void) { return 4; }

/* RAST/rasterize_points.cu:35-117: zero-filled outputs, P==0 early-out.  Returns the handle through *out_handle. */
int ref_forward(const Inputs* in, State* st, float* out_color, float* out_depth, int* out_radii, void** out_handle) {
    Handle* h = new Handle;
    h->cfg = *in;
    const int P = in->P, W = in->W, H = in->H, M = in->M;
    const size_t N = (size_t)W * H;
    float* d_color = h->zeros<float>(3 * N);
    float* d_depth = h->zeros<float>(N);
    h->radii = h->zeros<int>(P);
    h->bg = h->up(in->background, 3);
    h->means3D = h->up(in->means3D, (size_t)P * 3);
    h->shs = h->up(in->shs, (size_t)P * M * 3);
    h->colors = h->up(in->colors_precomp, (size_t)P * 3);
    h->opac = h->up(in->opacities, P);
    h->scales = h->up(in->scales, (size_t)P * 3);
    h->rots = h->up(in->rotations, (size_t)P * 4);
    h->cov = h->up(in->cov3D_precomp, (size_t)P * 6);
    h->view = h->up(in->viewmatrix, 16);
    h->proj = h->up(in->projmatrix, 16);
    h->campos = h->up(in->cam_pos, 3);
    int R = 0;
    if (P != 0) {
        R = CudaRasterizer::Rasterizer::forward(h->geom.fn(), h->binning.fn(), h->img.fn(), P, in->D, M, h->bg, W, H,
                                                h->means3D, h->shs, h->colors, h->opac, h->scales, in->scale_modifier,
                                                h->rots, h->cov, h->view, h->proj, h->campos, in->tan_fovx,
                                                in->tan_fovy, false, d_color, d_depth, h->radii, false);
        HIPCHK(hipDeviceSynchronize());
    }
    h->R = R;
    down(out_color, d_color, 3 * N);
    down(out_depth, d_depth, N);
    down(out_radii, h->radii, (size_t)P);
    st->num_rendered = R;
    st->point_list_keys = nullptr;
    st->point_list = nullptr;
    if (P != 0) {
        char* c = h->geom.ptr;
        CudaRasterizer::GeometryState g = CudaRasterizer::GeometryState::fromChunk(c, P);
        down(st->depths, g.depths, (size_t)P);
        down((bool*)st->clamped, g.clamped, (size_t)P * 3);
        down(st->means2D, (float*)g.means2D, (size_t)P * 2);
        down(st->cov3D, g.cov3D, (size_t)P * 6);
        down(st->conic_opacity, (float*)g.conic_opacity, (size_t)P * 4);
        down(st->rgb, g.rgb, (size_t)P * 3);
        down(st->tiles_touched, g.tiles_touched, (size_t)P);
        down(st->point_offsets, g.point_offsets, (size_t)P);
        c = h->img.ptr;
        CudaRasterizer::ImageState im = CudaRasterizer::ImageState::fromChunk(c, N);
        const size_t tiles = (size_t)((W + 15) / 16) * ((H + 15) / 16);
        down(st->final_T, im.accum_alpha, N);
        down(st->n_contrib, im.n_contrib, N);
        down(st->ranges, (uint32_t*)im.ranges, tiles * 2);
        if (R > 0) {
            c = h->binning.ptr;
            CudaRasterizer::BinningState b = CudaRasterizer::BinningState::fromChunk(c, R);
            st->point_list_keys = (uint64_t*)malloc((size_t)R * 8);
            st->point_list = (uint32_t*)malloc((size_t)R * 4);
            down(st->point_list_keys, b.point_list_keys, (size_t)R);
            down(st->point_list, b.point_list, (size_t)R);
        }
    }
    *out_handle = h;
    return R;
}

void ref_free_binning(State* st) {
    free(st->point_list_keys);
    free(st->point_list);
    st->point_list_keys = nullptr;
    st->point_list = nullptr;
}

/* RAST/rasterize_points.cu:119-202: ten zero-filled gradient arrays, then Rasterizer::backward. */
void ref_backward(void* handle, const float* dL_dout_color, const float* dL_dout_depth, float* dL_dmeans2D,
                  float* dL_dconic, float* dL_dopacity, float* dL_dcolors, float* dL_ddepths, float* dL_dmeans3D,
                  float* dL_dcov3D, float* dL_dsh, float* dL_dscales, float* dL_drotations) {
    Handle* h = (Handle*)handle;
    const Inputs& in = h->cfg;
    const int P = in.P, W = in.W, H = in.H, M = in.M;
    const size_t N = (size_t)W * H;
    if (P == 0) return;
    Handle tmp; /* owns the per-call device arrays */
    float* gcol = tmp.up(dL_dout_color, 3 * N);
    float* gdep = tmp.up(dL_dout_depth, N);
    float* d_m2 = tmp.zeros<float>((size_t)P * 3);
    float* d_con = tmp.zeros<float>((size_t)P * 4);
    float* d_op = tmp.zeros<float>(P);
    float* d_col = tmp.zeros<float>((size_t)P * 3);
    float* d_dep = tmp.zeros<float>(P);
    float* d_m3 = tmp.zeros<float>((size_t)P * 3);
    float* d_cov = tmp.zeros<float>((size_t)P * 6);
    float* d_sh = tmp.zeros<float>((size_t)P * M * 3);
    float* d_sc = tmp.zeros<float>((size_t)P * 3);
    float* d_rot = tmp.zeros<float>((size_t)P * 4);
    CudaRasterizer::Rasterizer::backward(P, in.D, M, h->R, h->bg, W, H, h->means3D, h->shs, h->colors, h->scales,
                                         in.scale_modifier, h->rots, h->cov, h->view, h->proj, h->campos, in.tan_fovx,
                                         in.tan_fovy, h->radii, h->geom.ptr, h->binning.ptr, h->img.ptr, gcol, gdep,
                                         d_m2, d_con, d_op, d_col, d_dep, d_m3, d_cov, d_sh, d_sc, d_rot, false);
    HIPCHK(hipDeviceSynchronize());
    down(dL_dmeans2D, d_m2, (size_t)P * 3);
    down(dL_dconic, d_con, (size_t)P * 4);
    down(dL_dopacity, d_op, (size_t)P);
    down(dL_dcolors, d_col, (size_t)P * 3);
    down(dL_ddepths, d_dep, (size_t)P);
    down(dL_dmeans3D, d_m3, (size_t)P * 3);
    down(dL_dcov3D, d_cov, (size_t)P * 6);
    down(dL_dsh, d_sh, (size_t)P * M * 3);
    down(dL_dscales, d_sc, (size_t)P * 3);
    down(dL_drotations, d_rot, (size_t)P * 4);
}

void ref_free(void* handle) { delete (Handle*)handle; }

/* RAST/rasterize_points.cu:204-223 */
void ref_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix, uint8_t* present) {
    if (P == 0) return;
    Handle tmp;
    float* m = tmp.up(means3D, (size_t)P * 3);
    float* v = tmp.up(viewmatrix, 16);
    float* p = tmp.up(projmatrix, 16);
    bool* out = tmp.zeros<bool>(P);
    CudaRasterizer::Rasterizer::markVisible(P, m, v, p, out);
    HIPCHK(hipDeviceSynchronize());
    down((bool*)present, out, (size_t)P);
}


/* ---- device-pointer entry points: RAST/rasterize_points.cu:35-202 without torch ------------------------------------------
 * Everything below takes DEVICE pointers and copies nothing: the three arenas come from caller callbacks (the reference's
 * std::function<char*(size_t)> resize functions, rasterize_points.cu:27-33 -- here plain C function pointers that a torch
 * caller backs with uint8 tensors), outputs are caller-allocated and caller-zeroed exactly like the `torch::full(..., 0.0)`
 * / `torch::zeros` of the reference.  oracle/ref_diff_raster_C.py binds these with the `_C` signatures of
 * RAST/rasterize_points.h:18-68, so that the reference's OWN Python wrapper (diff_gaussian_rasterization/__init__.py) and the
 * reference's OWN train.py run on the reference's OWN kernels on this GPU: the end-to-end oracle of the PSNR-parity test.
 * The kernels run on the legacy default stream (the reference launches everything there): callers synchronise. */
typedef char* (*ref_resize_fn)(void* user, size_t bytes);

int ref_forward_dev(const Inputs* in, ref_resize_fn geom, void* geom_user, ref_resize_fn binning, void* binning_user,
                    ref_resize_fn img, void* img_user, float* out_color, float* out_depth, int* radii, int prefiltered, int debug) {
    if (in->P == 0) return 0;
    std::function<char*(size_t)> g = [=](size_t n) { return geom(geom_user, n); };
    std::function<char*(size_t)> b = [=](size_t n) { return binning(binning_user, n); };
    std::function<char*(size_t)> i = [=](size_t n) { return img(img_user, n); };
    HIPCHK(hipDeviceSynchronize());   /* inputs were produced on the caller's stream */
    const int R = CudaRasterizer::Rasterizer::forward(g, b, i, in->P, in->D, in->M, in->background, in->W, in->H, in->means3D,
                                                      in->shs, in->colors_precomp, in->opacities, in->scales, in->scale_modifier,
                                                      in->rotations, in->cov3D_precomp, in->viewmatrix, in->projmatrix, in->cam_pos,
                                                      in->tan_fovx, in->tan_fovy, prefiltered != 0, out_color, out_depth, radii,
                                                      debug != 0);
    HIPCHK(hipDeviceSynchronize());
    return R;
}

void ref_backward_dev(const Inputs* in, int R, const int* radii, char* geom, char* binning, char* img, const float* dL_dout_color,
                      const float* dL_dout_depth, float* dL_dmeans2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolors,
                      float* dL_ddepths, float* dL_dmeans3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscales,
                      float* dL_drotations, int debug) {
    if (in->P == 0) return;
    HIPCHK(hipDeviceSynchronize());
    CudaRasterizer::Rasterizer::backward(in->P, in->D, in->M, R, in->background, in->W, in->H, in->means3D, in->shs,
                                         in->colors_precomp, in->scales, in->scale_modifier, in->rotations, in->cov3D_precomp,
                                         in->viewmatrix, in->projmatrix, in->cam_pos, in->tan_fovx, in->tan_fovy, radii, geom,
                                         binning, img, dL_dout_color, dL_dout_depth, dL_dmeans2D, dL_dconic, dL_dopacity,
                                         dL_dcolors, dL_ddepths, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations,
                                         debug != 0);
    HIPCHK(hipDeviceSynchronize());
}

void ref_mark_visible_dev(int P, const float* means3D, const float* viewmatrix, const float* projmatrix, uint8_t* present) {
    if (P == 0) return;
    HIPCHK(hipDeviceSynchronize());
    CudaRasterizer::Rasterizer::markVisible(P, (float*)means3D, (float*)viewmatrix, (float*)projmatrix, (bool*)present);
    HIPCHK(hipDeviceSynchronize());
}

}  // extern "C"
