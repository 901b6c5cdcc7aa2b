/*
 * raster_oracle.c -- CPU restatement of the neural-mesh-renderer rasteriser kernels.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the shipped package may import, link or call
 * this file; it is the checker for tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg.
 *
 * PARITY UNPINNED: the arithmetic lives in the third-party `neural_renderer` package
 * (pip `neural-renderer-pytorch`, un-pinned in /root/reference/environment.yml:36;
 * upstream daniilidis-group/neural_renderer, neural_renderer/cuda/rasterize_cuda_kernel.cu),
 * which is NOT present under /root/reference nor anywhere in this image, and the
 * reference ships no test / golden vector for it.  The functions below restate the
 * published algorithm of that package's six CUDA kernels; they are anchored on the
 * reference's own call sites (meshreg/neurender/rasterize.py:202-315: argument order,
 * buffer pre-fill values, which outputs exist) and pinned by the hand-derived
 * known-answer tests in tests/test_oracle_raster.py.
 *
 * Arithmetic: plain fp32, one IEEE operation per C operator (compile with
 * -ffp-contract=off, no -ffast-math).  Where upstream mixes double literals into fp32
 * expressions ("2. * yi", "1. / x", "0.5 * x") the double op is exact or its double
 * rounding is innocuous for fp32 (53 >= 2*24+2), so the fp32 form used here gives the
 * same bits.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define ORACLE_API __attribute__((visibility("default")))

static inline float fmin2(float a, float b) { return fminf(a, b); }
static inline float fmax2(float a, float b) { return fmaxf(a, b); }

static inline int backfacing(const float* f) {
    /* kernel A/B/D "return if backside" */
    return (f[7] - f[1]) * (f[3] - f[0]) < (f[4] - f[1]) * (f[6] - f[0]);
}

/* Kernel A: per-face pixel-space inverse.  rasterize.py:202 (first launch of
 * forward_face_index_map).  Back-facing faces leave faces_inv untouched. */
static void face_inverse(const float* face, float* face_inv, int is) {
    float p[3][2];
    for (int num = 0; num < 3; num++)
        for (int dim = 0; dim < 2; dim++)
            p[num][dim] = 0.5f * (face[3 * num + dim] * (float)is + (float)is - 1.0f);
    float inv[9] = {p[1][1] - p[2][1], p[2][0] - p[1][0], p[1][0] * p[2][1] - p[2][0] * p[1][1],
                    p[2][1] - p[0][1], p[0][0] - p[2][0], p[2][0] * p[0][1] - p[0][0] * p[2][1],
                    p[0][1] - p[1][1], p[1][0] - p[0][0], p[0][0] * p[1][1] - p[1][0] * p[0][1]};
    float den = (p[2][0] * (p[0][1] - p[1][1]) + p[0][0] * (p[1][1] - p[2][1]) +
                 p[1][0] * (p[2][1] - p[0][1]));
    for (int k = 0; k < 9; k++) face_inv[k] = inv[k] / den;
}

/* Kernels A + B.  rasterize.py:202-215.  Buffers pre-filled by the caller exactly as
 * rasterize.py:60-85 does (face_index_map=-1, weight_map=0, depth_map=far, face_inv_map=0,
 * faces_inv=0). */
ORACLE_API void oracle_forward_face_index_map(const float* faces, int32_t* face_index_map,
                                              float* weight_map, float* depth_map,
                                              float* face_inv_map, float* faces_inv,
                                              int batch_size, int num_faces, int image_size,
                                              float near_, float far_, int return_rgb,
                                              int return_alpha, int return_depth,
                                              int num_threads) {
    (void)return_rgb;
    (void)return_alpha;
    const int is = image_size, nf = num_faces;
    for (int64_t i = 0; i < (int64_t)batch_size * nf; i++) {
        const float* face = &faces[i * 9];
        if (backfacing(face)) continue;
        face_inverse(face, &faces_inv[i * 9], is);
    }
#ifdef _OPENMP
#pragma omp parallel for num_threads(num_threads > 0 ? num_threads : 1) schedule(dynamic, 4) collapse(2)
#endif
    for (int bn = 0; bn < batch_size; bn++) {
        for (int yi = 0; yi < is; yi++) {
            for (int xi = 0; xi < is; xi++) {
                const int64_t i = ((int64_t)bn * is + yi) * is + xi;
                const float yp = (float)(2 * yi + 1 - is) / (float)is;
                const float xp = (float)(2 * xi + 1 - is) / (float)is;
                float depth_min = far_;
                int face_index_min = -1;
                float weight_min[3] = {0, 0, 0};
                const float* inv_min = 0;
                for (int fn = 0; fn < nf; fn++) {
                    const float* face = &faces[((int64_t)bn * nf + fn) * 9];
                    const float* face_inv = &faces_inv[((int64_t)bn * nf + fn) * 9];
                    if (backfacing(face)) continue;
                    if (((yp - face[1]) * (face[3] - face[0]) < (xp - face[0]) * (face[4] - face[1])) ||
                        ((yp - face[4]) * (face[6] - face[3]) < (xp - face[3]) * (face[7] - face[4])) ||
                        ((yp - face[7]) * (face[0] - face[6]) < (xp - face[6]) * (face[1] - face[7])))
                        continue;
                    float w[3];
                    w[0] = face_inv[0] * (float)xi + face_inv[1] * (float)yi + face_inv[2];
                    w[1] = face_inv[3] * (float)xi + face_inv[4] * (float)yi + face_inv[5];
                    w[2] = face_inv[6] * (float)xi + face_inv[7] * (float)yi + face_inv[8];
                    float w_sum = 0;
                    for (int k = 0; k < 3; k++) {
                        w[k] = fmin2(fmax2(w[k], 0.0f), 1.0f);
                        w_sum += w[k];
                    }
                    for (int k = 0; k < 3; k++) w[k] /= w_sum;
                    const float zp = 1.0f / (w[0] / face[2] + w[1] / face[5] + w[2] / face[8]);
                    if (zp <= near_ || far_ <= zp) continue;
                    if (zp < depth_min) {
                        depth_min = zp;
                        face_index_min = fn;
                        for (int k = 0; k < 3; k++) weight_min[k] = w[k];
                        inv_min = face_inv;
                    }
                }
                if (0 <= face_index_min) {
                    depth_map[i] = depth_min;
                    face_index_map[i] = face_index_min;
                    for (int k = 0; k < 3; k++) weight_map[3 * i + k] = weight_min[k];
                    if (return_depth)
                        for (int k = 0; k < 9; k++) face_inv_map[9 * i + k] = inv_min[k];
                }
            }
        }
    }
}

/* Kernel C.  rasterize.py:232-243. */
ORACLE_API void oracle_forward_texture_sampling(const float* faces, const float* textures,
                                                const int32_t* face_index_map,
                                                const float* weight_map, const float* depth_map,
                                                float* rgb_map, int32_t* sampling_index_map,
                                                float* sampling_weight_map, int batch_size,
                                                int num_faces, int image_size, int texture_size,
                                                float eps) {
    const int is = image_size, nf = num_faces, ts = texture_size;
    const int64_t npx = (int64_t)batch_size * is * is;
    for (int64_t i = 0; i < npx; i++) {
        const int face_index = face_index_map[i];
        if (face_index < 0) continue;
        const int64_t bn = i / ((int64_t)is * is);
        const float* face = &faces[(bn * nf + face_index) * 9];
        const float* texture = &textures[(bn * nf + face_index) * ts * ts * ts * 3];
        const float* weight = &weight_map[i * 3];
        const float depth = depth_map[i];
        float tif[3];
        for (int k = 0; k < 3; k++) {
            float t = weight[k] * (float)(ts - 1) * (depth / face[3 * k + 2]);
            t = fmax2(t, 0.0f);
            t = fmin2(t, (float)(ts - 1) - eps);
            tif[k] = t;
        }
        float new_pixel[3] = {0, 0, 0};
        for (int pn = 0; pn < 8; pn++) {
            float w = 1;
            int tii[3];
            for (int k = 0; k < 3; k++) {
                if (((pn >> k) % 2) == 0) {
                    w *= 1.0f - (tif[k] - (float)(int)tif[k]);
                    tii[k] = (int)tif[k];
                } else {
                    w *= tif[k] - (float)(int)tif[k];
                    tii[k] = (int)tif[k] + 1;
                }
            }
            const int isc = tii[0] * ts * ts + tii[1] * ts + tii[2];
            for (int k = 0; k < 3; k++) new_pixel[k] += w * texture[isc * 3 + k];
            sampling_index_map[i * 8 + pn] = isc;
            sampling_weight_map[i * 8 + pn] = w;
        }
        for (int k = 0; k < 3; k++) rgb_map[i * 3 + k] = new_pixel[k];
    }
}

/* Kernel D: NMR pseudo-gradient of rgb/alpha w.r.t. the x,y of the face vertices.
 * rasterize.py:269-281.  One serial walk per face, three edges x two axes. */
ORACLE_API void oracle_backward_pixel_map(const float* faces, const int32_t* face_index_map,
                                          const float* rgb_map, const float* alpha_map,
                                          const float* grad_rgb_map,
                                          const float* grad_alpha_map, float* grad_faces,
                                          int batch_size, int num_faces, int image_size,
                                          float eps, int return_rgb, int return_alpha,
                                          int num_threads) {
    const int is = image_size;
    const float fis = (float)is;
#ifdef _OPENMP
#pragma omp parallel for num_threads(num_threads > 0 ? num_threads : 1) schedule(dynamic, 64)
#endif
    for (int64_t i = 0; i < (int64_t)batch_size * num_faces; i++) {
        const int64_t bn = i / num_faces;
        const int fn = (int)(i % num_faces);
        const float* face = &faces[i * 9];
        float grad_face[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (backfacing(face)) continue;

        for (int edge_num = 0; edge_num < 3; edge_num++) {
            int pi[3];
            float pp[3][2];
            for (int num = 0; num < 3; num++) pi[num] = (edge_num + num) % 3;
            for (int num = 0; num < 3; num++)
                for (int dim = 0; dim < 2; dim++)
                    pp[num][dim] = 0.5f * (face[3 * pi[num] + dim] * fis + fis - 1.0f);

            for (int axis = 0; axis < 2; axis++) {
                float p[3][2];
                for (int num = 0; num < 3; num++)
                    for (int dim = 0; dim < 2; dim++) p[num][dim] = pp[num][(dim + axis) % 2];

                int direction;
                if (axis == 0)
                    direction = (p[0][0] < p[1][0]) ? -1 : 1;
                else
                    direction = (p[0][0] < p[1][0]) ? 1 : -1;

                /* int d0_from = max(ceil(min(p0,p1)), 0.); int d0_to = min(max(p0,p1), is-1.) */
                const int d0_from = (int)fmax2(ceilf(fmin2(p[0][0], p[1][0])), 0.0f);
                const int d0_to = (int)fmin2(fmax2(p[0][0], p[1][0]), fis - 1.0f);
                for (int d0 = d0_from; d0 <= d0_to; d0++) {
                    const float d1_cross =
                        (p[1][1] - p[0][1]) / (p[1][0] - p[0][0]) * ((float)d0 - p[0][0]) + p[0][1];
                    int d1_in;
                    if (0 < direction)
                        d1_in = (int)floorf(d1_cross);
                    else
                        d1_in = (int)ceilf(d1_cross);
                    const int d1_out = d1_in + direction;
                    if (d1_in < 0 || is <= d1_in) continue;
                    if (d1_out < 0 || is <= d1_out) continue;

                    float alpha_in = 0, alpha_out = 0;
                    const float *rgb_in = 0, *rgb_out = 0;
                    int64_t map_index_in, map_index_out;
                    if (axis == 0) {
                        map_index_in = bn * is * is + (int64_t)d1_in * is + d0;
                        map_index_out = bn * is * is + (int64_t)d1_out * is + d0;
                    } else {
                        map_index_in = bn * is * is + (int64_t)d0 * is + d1_in;
                        map_index_out = bn * is * is + (int64_t)d0 * is + d1_out;
                    }
                    if (return_alpha) {
                        alpha_in = alpha_map[map_index_in];
                        alpha_out = alpha_map[map_index_out];
                    }
                    if (return_rgb) {
                        rgb_in = &rgb_map[map_index_in * 3];
                        rgb_out = &rgb_map[map_index_out * 3];
                    }
                    const int64_t map_offset = (axis == 0) ? is : 1;

                    /* out: pixels beyond the edge, walking away from the face */
                    if (face_index_map[map_index_in] == fn) {
                        const int d1_limit = (0 < direction) ? is - 1 : 0;
                        int d1_from = d1_out < d1_limit ? d1_out : d1_limit;
                        if (d1_from < 0) d1_from = 0;
                        int d1_to = d1_out > d1_limit ? d1_out : d1_limit;
                        if (d1_to > is - 1) d1_to = is - 1;
                        int64_t mi = (axis == 0) ? bn * is * is + (int64_t)d1_from * is + d0
                                                 : bn * is * is + (int64_t)d0 * is + d1_from;
                        for (int d1 = d1_from; d1 <= d1_to; d1++, mi += map_offset) {
                            float diff_grad = 0;
                            if (return_alpha)
                                diff_grad += (alpha_map[mi] - alpha_in) * grad_alpha_map[mi];
                            if (return_rgb)
                                for (int k = 0; k < 3; k++)
                                    diff_grad += (rgb_map[mi * 3 + k] - rgb_in[k]) * grad_rgb_map[mi * 3 + k];
                            if (diff_grad <= 0) continue;
                            if (p[1][0] != (float)d0) {
                                float dist = (p[1][0] - p[0][0]) / (p[1][0] - (float)d0) *
                                             ((float)d1 - d1_cross) * 2.0f / fis;
                                dist = (0 < dist) ? dist + eps : dist - eps;
                                grad_face[pi[0] * 3 + (1 - axis)] -= diff_grad / dist;
                            }
                            if (p[0][0] != (float)d0) {
                                float dist = (p[1][0] - p[0][0]) / ((float)d0 - p[0][0]) *
                                             ((float)d1 - d1_cross) * 2.0f / fis;
                                dist = (0 < dist) ? dist + eps : dist - eps;
                                grad_face[pi[1] * 3 + (1 - axis)] -= diff_grad / dist;
                            }
                        }
                    }

                    /* in: pixels of this face between the edge and the opposite boundary */
                    {
                        float d0_cross2;
                        if (((float)d0 - p[0][0]) * ((float)d0 - p[2][0]) < 0)
                            d0_cross2 = (p[2][1] - p[0][1]) / (p[2][0] - p[0][0]) * ((float)d0 - p[0][0]) + p[0][1];
                        else
                            d0_cross2 = (p[1][1] - p[2][1]) / (p[1][0] - p[2][0]) * ((float)d0 - p[2][0]) + p[2][1];
                        int d1_limit;
                        if (0 < direction)
                            d1_limit = (int)ceilf(d0_cross2);
                        else
                            d1_limit = (int)floorf(d0_cross2);
                        int d1_from = d1_in < d1_limit ? d1_in : d1_limit;
                        if (d1_from < 0) d1_from = 0;
                        int d1_to = d1_in > d1_limit ? d1_in : d1_limit;
                        if (d1_to > is - 1) d1_to = is - 1;
                        int64_t mi = (axis == 0) ? bn * is * is + (int64_t)d1_from * is + d0
                                                 : bn * is * is + (int64_t)d0 * is + d1_from;
                        for (int d1 = d1_from; d1 <= d1_to; d1++, mi += map_offset) {
                            if (face_index_map[mi] != fn) continue;
                            float diff_grad = 0;
                            if (return_alpha)
                                diff_grad += (alpha_map[mi] - alpha_out) * grad_alpha_map[mi];
                            if (return_rgb)
                                for (int k = 0; k < 3; k++)
                                    diff_grad += (rgb_map[mi * 3 + k] - rgb_out[k]) * grad_rgb_map[mi * 3 + k];
                            if (diff_grad <= 0) continue;
                            if (p[1][0] != (float)d0) {
                                float dist = (p[1][0] - p[0][0]) / (p[1][0] - (float)d0) *
                                             ((float)d1 - d1_cross) * 2.0f / fis;
                                dist = (0 < dist) ? dist + eps : dist - eps;
                                grad_face[pi[0] * 3 + (1 - axis)] -= diff_grad / dist;
                            }
                            if (p[0][0] != (float)d0) {
                                float dist = (p[1][0] - p[0][0]) / ((float)d0 - p[0][0]) *
                                             ((float)d1 - d1_cross) * 2.0f / fis;
                                dist = (0 < dist) ? dist + eps : dist - eps;
                                grad_face[pi[1] * 3 + (1 - axis)] -= diff_grad / dist;
                            }
                        }
                    }
                }
            }
        }
        for (int k = 0; k < 9; k++) grad_faces[i * 9 + k] = grad_face[k];
    }
}

/* Kernel E: adjoint of the texture sampling.  rasterize.py:290-297.  Serial pixel order
 * (the upstream atomicAdd order is unspecified). */
ORACLE_API void oracle_backward_textures(const int32_t* face_index_map,
                                         const float* sampling_weight_map,
                                         const int32_t* sampling_index_map,
                                         const float* grad_rgb_map, float* grad_textures,
                                         int batch_size, int num_faces, int image_size,
                                         int texture_size) {
    const int is = image_size, nf = num_faces, ts = texture_size;
    const int64_t npx = (int64_t)batch_size * is * is;
    for (int64_t i = 0; i < npx; i++) {
        const int face_index = face_index_map[i];
        if (face_index < 0) continue;
        const int64_t bn = i / ((int64_t)is * is);
        float* grad_texture = &grad_textures[(bn * nf + face_index) * ts * ts * ts * 3];
        for (int pn = 0; pn < 8; pn++) {
            const float w = sampling_weight_map[i * 8 + pn];
            const int isc = sampling_index_map[i * 8 + pn];
            for (int k = 0; k < 3; k++) grad_texture[isc * 3 + k] += w * grad_rgb_map[i * 3 + k];
        }
    }
}

/* Kernel F: analytic gradient of the depth map.  rasterize.py:306-315. */
ORACLE_API void oracle_backward_depth_map(const float* faces, const float* depth_map,
                                          const int32_t* face_index_map,
                                          const float* face_inv_map, const float* weight_map,
                                          const float* grad_depth_map, float* grad_faces,
                                          int batch_size, int num_faces, int image_size) {
    const int is = image_size, nf = num_faces;
    const int64_t npx = (int64_t)batch_size * is * is;
    for (int64_t i = 0; i < npx; i++) {
        const int fn = face_index_map[i];
        if (fn < 0) continue;
        const int64_t bn = i / ((int64_t)is * is);
        const float* face = &faces[(bn * nf + fn) * 9];
        const float depth = depth_map[i];
        const float depth2 = depth * depth;
        const float* face_inv = &face_inv_map[i * 9];
        const float* weight = &weight_map[i * 3];
        const float grad_depth = grad_depth_map[i];
        float* grad_face = &grad_faces[(bn * nf + fn) * 9];
        for (int k = 0; k < 3; k++) {
            const float z_k = face[3 * k + 2];
            grad_face[3 * k + 2] += grad_depth * weight[k] * depth2 / (z_k * z_k);
        }
        float tmp[3] = {0, 0, 0};
        for (int k = 0; k < 3; k++)
            for (int l = 0; l < 3; l++) tmp[k] += -face_inv[3 * l + k] / face[3 * l + 2];
        for (int k = 0; k < 3; k++)
            for (int l = 0; l < 2; l++)
                grad_face[3 * k + l] += -grad_depth * tmp[l] * weight[k] * depth2 * (float)is / 2.0f;
    }
}

ORACLE_API int oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
