// oracle/eigen_probe.cpp -- TEST INFRASTRUCTURE ONLY (builds into oracle/_ref/, git-ignored).
//
// The reference's hot-path .cpp files cannot be compiled in this image (they include
// <nanobind/...>, which is absent, and no stand-in headers are written).  What CAN be
// compiled straight from the read-only mount are the reference headers that do not touch
// nanobind -- rteigentypes.h, rtray.h, rthitrecord.h, rtmathutils.h -- and the reference's
// vendored Eigen (cpp/Eigen, 5.0.0-dev).  This probe includes exactly those, from where
// they lie, and evaluates
//   (1) the reference's own random_double() (rtmathutils.h:15-19), and
//   (2) Eigen expressions of the same *types and shapes* as the call sites on the hot path
//       (rtrender.cpp:24,32,54-58; rtrayintersection.cpp:30-32,42-43,80,91-93,101,105,120,127,173)
// on seeded inputs, and dumps inputs + results as raw little-endian arrays.  The outputs
// pin the floating-point evaluation order that oracle/rt_oracle.c restates in plain C
// (tests/test_oracle_pins.py compares bit patterns).  oracle/make_golden.py packs the
// dump into tests/golden/eigen_probe.npz so the check also runs where the mount is absent.
//
// Build: see oracle/Makefile (target `ref`).  Usage: eigen_probe <out_dir>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <limits>
#include <random>
#include <string>
#include <vector>

#include "rteigentypes.h"
#include "rthitrecord.h"
#include "rtmathutils.h"
#include "rtray.h"

namespace {

std::string g_dir;

template <typename T>
void dump(const char* name, const std::vector<T>& v) {
    std::string path = g_dir + "/" + name;
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) {
        std::perror(path.c_str());
        std::exit(2);
    }
    std::fwrite(v.data(), sizeof(T), v.size(), f);
    std::fclose(f);
}

// input generator (independent of the reference's RNG): doubles in [lo, hi)
struct Gen {
    std::mt19937_64 g{0xC0FFEEull};
    double uni(double lo, double hi) {
        double u = double(g() >> 11) * (1.0 / 9007199254740992.0);
        return lo + (hi - lo) * u;
    }
    int integer(int lo, int hi) { return lo + int(g() % uint64_t(hi - lo)); }
};

// Row-wise cross product written with the same Eigen column expressions the path uses.
EiVectorD3d rows_cross(const EiVectorD3d& a, const EiVectorD3d& b) {
    EiVectorD3d r(a.rows(), 3);
    r.col(0) = a.col(1).cwiseProduct(b.col(2)) - a.col(2).cwiseProduct(b.col(1));
    r.col(1) = a.col(2).cwiseProduct(b.col(0)) - a.col(0).cwiseProduct(b.col(2));
    r.col(2) = a.col(0).cwiseProduct(b.col(1)) - a.col(1).cwiseProduct(b.col(0));
    return r;
}

}  // namespace

int main(int argc, char** argv) {
    if (argc < 2) {
        std::fprintf(stderr, "usage: %s <out_dir>\n", argv[0]);
        return 2;
    }
    g_dir = argv[1];
    Gen gen;

    // ---- (1) the reference's random_double(), fresh process => mt19937(5489) position 0 ----
    {
        std::vector<double> r(6000);
        for (auto& x : r) x = random_double();
        dump("rng_doubles.f64", r);
        std::vector<uint32_t> w(2000);  // raw words that follow (positions 12000..13999)
        for (auto& x : w) x = uint32_t(generator());
        dump("rng_words_after.u32", w);
    }

    // ---- (2a) normalized() on 1x3 row vectors (rtrender.cpp:58) ----
    {
        const int n = 1200;
        std::vector<double> in(3 * n), out(3 * n);
        for (int i = 0; i < n; ++i) {
            EiVector3d v;
            double scale = (i % 7 == 0) ? 1e-3 : (i % 11 == 0 ? 1e4 : 1.0);
            v << gen.uni(-1, 1) * scale, gen.uni(-1, 1) * scale, gen.uni(-1, 1) * scale;
            if (i == 0) v << 0.0, 0.0, 0.0;
            if (i == 1) v << 0.0, -2.5, 0.0;
            EiVector3d u = v.normalized();
            for (int c = 0; c < 3; ++c) {
                in[3 * i + c] = v(c);
                out[3 * i + c] = u(c);
            }
        }
        dump("norm_in.f64", in);
        dump("norm_out.f64", out);
    }

    // ---- (2b) pixel sample + direction (rtrender.cpp:53-58) ----
    {
        const int n = 1200;
        std::vector<double> cam(3 * n), p00(3 * n), sp(6 * n), off(2 * n), ps_out(3 * n), dir_out(3 * n);
        std::vector<int32_t> ij(2 * n);
        for (int q = 0; q < n; ++q) {
            EiVector3d camera_center, pixel_00_center;
            Eigen::Matrix<double, 2, 3, Eigen::StorageOptions::RowMajor> spacing;
            camera_center << gen.uni(-2, 2), gen.uni(-2, 2), gen.uni(-2, 2);
            pixel_00_center << gen.uni(-4, 4), gen.uni(-4, 4), gen.uni(-4, -1);
            spacing << gen.uni(-0.03, 0.03), gen.uni(-0.03, 0.03), gen.uni(-0.03, 0.03), gen.uni(-0.03, 0.03),
                gen.uni(-0.03, 0.03), gen.uni(-0.03, 0.03);
            int i = gen.integer(0, 4096), j = gen.integer(0, 4096);
            double offset[2] = {gen.uni(-0.5, 0.5), gen.uni(-0.5, 0.5)};
            EiVector3d pixel_sample =
                pixel_00_center + (i + offset[0]) * spacing.row(0) + (j + offset[1]) * spacing.row(1);
            EiVector3d ray_direction = pixel_sample - camera_center;
            Ray ray{camera_center, ray_direction.normalized()};
            for (int c = 0; c < 3; ++c) {
                cam[3 * q + c] = camera_center(c);
                p00[3 * q + c] = pixel_00_center(c);
                ps_out[3 * q + c] = pixel_sample(c);
                dir_out[3 * q + c] = ray.direction(c);
            }
            for (int c = 0; c < 6; ++c) sp[6 * q + c] = spacing(c / 3, c % 3);
            off[2 * q] = offset[0];
            off[2 * q + 1] = offset[1];
            ij[2 * q] = i;
            ij[2 * q + 1] = j;
        }
        dump("ps_cam.f64", cam);
        dump("ps_p00.f64", p00);
        dump("ps_spacing.f64", sp);
        dump("ps_off.f64", off);
        dump("ps_ij.i32", ij);
        dump("ps_sample.f64", ps_out);
        dump("ps_dir.f64", dir_out);
    }

    // ---- (2c) Moller-Trumbore quantities on leaf-sized batches, with the path's Eigen types ----
    {
        const int n_batches = 1000;
        std::vector<double> o_in, d_in, tri_in, det_out, u_out, v_out, w_out, t_out;
        std::vector<int32_t> cnt_in, argmin_out;
        for (int b = 0; b < n_batches; ++b) {
            const unsigned int n = (unsigned int)gen.integer(1, 7);
            Ray ray;
            ray.origin << gen.uni(-1, 1), gen.uni(-1, 1), gen.uni(2, 4);
            EiVector3d dd;
            dd << gen.uni(-0.4, 0.4), gen.uni(-0.4, 0.4), -1.0;
            ray.direction = dd.normalized();
            ray.t_min = 0.0;
            std::vector<double> node_coords(9 * n);
            for (unsigned int k = 0; k < n; ++k) {
                // triangles scattered around a point on the ray so that a good share is hit
                double t0 = gen.uni(1.0, 5.0), s = gen.uni(0.2, 2.0);
                double cx = ray.origin(0) + t0 * ray.direction(0) + 0.4 * s * gen.uni(-1, 1);
                double cy = ray.origin(1) + t0 * ray.direction(1) + 0.4 * s * gen.uni(-1, 1);
                double cz = ray.origin(2) + t0 * ray.direction(2);
                for (int vtx = 0; vtx < 3; ++vtx) {
                    node_coords[9 * k + 3 * vtx + 0] = cx + s * gen.uni(-1, 1);
                    node_coords[9 * k + 3 * vtx + 1] = cy + s * gen.uni(-1, 1);
                    node_coords[9 * k + 3 * vtx + 2] = cz + 0.2 * s * gen.uni(-1, 1);
                }
                if (b % 2 == 0 && k > 0)  // exact duplicates inside a leaf: ties for minCoeff
                    for (int c = 0; c < 9; ++c) node_coords[9 * k + c] = node_coords[c];
            }
            // same container types as rtrayintersection.cpp:42-54
            EiVectorD3d ray_directions = ray.direction.replicate(n, 1);
            EiArrayD3d ray_origins = ray.origin.replicate(n, 1).array();
            EiMatrixDd edge0(n, 3), nEdge2(n, 3);
            Eigen::Array<double, Eigen::Dynamic, Eigen::Dynamic, Eigen::RowMajor> nodes0(n, 3);
            for (unsigned int k = 0; k < n; ++k)
                for (int j = 0; j < 3; ++j) {
                    edge0(k, j) = node_coords[(3 * k + 1) * 3 + j] - node_coords[(3 * k) * 3 + j];
                    nodes0(k, j) = node_coords[(3 * k) * 3 + j];
                    nEdge2(k, j) = node_coords[(3 * k + 2) * 3 + j] - node_coords[(3 * k) * 3 + j];
                }
            EiArrayD3d p_vec = rows_cross(ray_directions, nEdge2);
            Eigen::Array<double, Eigen::Dynamic, 1> determinants = (edge0.array() * p_vec).rowwise().sum();
            Eigen::Array<double, Eigen::Dynamic, 1> inverse_determinants = determinants.inverse();
            EiArrayD3d t_vec = ray_origins - nodes0;
            Eigen::Array<double, Eigen::Dynamic, 1> bu =
                ((t_vec * p_vec).rowwise().sum()).array() * inverse_determinants;
            EiArrayD3d q_vec = rows_cross(t_vec.matrix(), edge0);
            Eigen::Array<double, Eigen::Dynamic, 1> bv =
                (ray_directions.array() * q_vec).rowwise().sum().matrix().array() * inverse_determinants;
            Eigen::Array<double, Eigen::Dynamic, 1> tv =
                (nEdge2.array() * q_vec).rowwise().sum().array() * inverse_determinants;
            Eigen::ArrayXXd bary(n, 3);
            bary.col(0) = bu;
            bary.col(1) = bv;
            bary.col(2) = 1.0 - bu - bv;
            // masked t values (mask logic is plain C++ comparisons on the arrays above)
            Eigen::Array<double, Eigen::Dynamic, 1> masked = tv;
            for (unsigned int k = 0; k < n; ++k) {
                bool ok = (determinants(k) > 1e-6) && (determinants(k) > 0) && (bu(k) >= 0) && (bu(k) <= 1) &&
                          (bv(k) >= 0) && ((bu(k) + bv(k)) <= 1) && (tv(k) >= ray.t_min) && (tv(k) <= ray.t_max);
                if (!ok) masked(k) = std::numeric_limits<double>::infinity();
            }
            Eigen::Index min_row, min_col;
            masked.minCoeff(&min_row, &min_col);

            cnt_in.push_back(int32_t(n));
            argmin_out.push_back(int32_t(min_row));
            for (int c = 0; c < 3; ++c) {
                o_in.push_back(ray.origin(c));
                d_in.push_back(ray.direction(c));
            }
            for (double x : node_coords) tri_in.push_back(x);
            for (unsigned int k = 0; k < n; ++k) {
                det_out.push_back(determinants(k));
                u_out.push_back(bary(k, 0));
                v_out.push_back(bary(k, 1));
                w_out.push_back(bary(k, 2));
                t_out.push_back(masked(k));
            }
        }
        dump("mt_count.i32", cnt_in);
        dump("mt_origin.f64", o_in);
        dump("mt_dir.f64", d_in);
        dump("mt_tris.f64", tri_in);
        dump("mt_det.f64", det_out);
        dump("mt_u.f64", u_out);
        dump("mt_v.f64", v_out);
        dump("mt_w.f64", w_out);
        dump("mt_t.f64", t_out);
        dump("mt_argmin.i32", argmin_out);
    }

    // ---- (2d) slab test: Eigen 1/array + std::min/std::max composition (rtrayintersection.cpp:124-144) ----
    {
        const int n = 3000;
        std::vector<double> o_in(3 * n), d_in(3 * n), mn_in(3 * n), mx_in(3 * n), tmin_out(n), tmax_out(n);
        std::vector<int32_t> hit_out(n);
        for (int q = 0; q < n; ++q) {
            Ray ray;
            double mn[3], mx[3];
            const bool degenerate = (q % 3 == 0);  // integer lattice, zero direction components -> 0*inf = NaN
            for (int c = 0; c < 3; ++c) {
                if (degenerate) {
                    ray.origin(c) = double(gen.integer(-2, 3));
                    int dc = gen.integer(-1, 2);
                    ray.direction(c) = (dc == 0 && gen.integer(0, 2)) ? -0.0 : double(dc);
                    int a = gen.integer(-2, 3), b2 = gen.integer(-2, 3);
                    mn[c] = double(std::min(a, b2));
                    mx[c] = double(std::max(a, b2));
                } else {
                    ray.origin(c) = gen.uni(-3, 3);
                    ray.direction(c) = gen.uni(-1, 1);
                    double a = gen.uni(-2, 2), b2 = gen.uni(-2, 2);
                    mn[c] = std::min(a, b2);
                    mx[c] = std::max(a, b2);
                }
            }
            ray.t_min = 0.0;
            double t_axis[6];
            EiVector3d inverse_direction = 1 / (ray.direction.array());
            for (int i = 0; i < 3; ++i) {
                t_axis[2 * i] = (mn[i] - ray.origin(i)) * inverse_direction(i);
                t_axis[2 * i + 1] = (mx[i] - ray.origin(i)) * inverse_direction(i);
            }
            double t_min = std::max(std::max(std::min(t_axis[0], t_axis[1]), std::min(t_axis[2], t_axis[3])),
                                    std::min(t_axis[4], t_axis[5]));
            double t_max = std::min(std::min(std::max(t_axis[0], t_axis[1]), std::max(t_axis[2], t_axis[3])),
                                    std::max(t_axis[4], t_axis[5]));
            bool hit = t_min <= t_max && t_max > 0.0 && t_min < ray.t_max;
            for (int c = 0; c < 3; ++c) {
                o_in[3 * q + c] = ray.origin(c);
                d_in[3 * q + c] = ray.direction(c);
                mn_in[3 * q + c] = mn[c];
                mx_in[3 * q + c] = mx[c];
            }
            tmin_out[q] = t_min;
            tmax_out[q] = t_max;
            hit_out[q] = hit ? 1 : 0;
        }
        dump("slab_origin.f64", o_in);
        dump("slab_dir.f64", d_in);
        dump("slab_min.f64", mn_in);
        dump("slab_max.f64", mx_in);
        dump("slab_tmin.f64", tmin_out);
        dump("slab_tmax.f64", tmax_out);
        dump("slab_hit.i32", hit_out);
    }

    // ---- (2e) shading expressions (rtrender.cpp:24, :28-32) and luma/quantise (:61-65) ----
    {
        const int n = 1200;
        std::vector<double> bary_in(3 * n), fc_in(3 * n), hit_out(3 * n), diry_in(n), sky_out(3 * n);
        std::vector<double> pc_in(3 * n), inten_out(n);
        std::vector<int32_t> spp_in(n), byte_out(n);
        for (int q = 0; q < n; ++q) {
            HitRecord rec;
            double u = gen.uni(0, 1), v = gen.uni(0, 1 - u);
            rec.barycentric_coordinates << u, v, 1.0 - u - v;
            double g = gen.uni(0, 1);
            if (q % 2)
                rec.face_color << g, g, g;
            else
                rec.face_color << gen.uni(0, 1), gen.uni(0, 1), gen.uni(0, 1);
            EiVector3d c = rec.barycentric_coordinates(0) * rec.face_color +
                           rec.barycentric_coordinates(1) * rec.face_color +
                           rec.barycentric_coordinates(2) * rec.face_color;
            double dir_y = gen.uni(-1, 1);
            double a = 0.5 * (dir_y + 1.0);
            EiVector3d white, blue;
            white << 1.0, 1.0, 1.0;
            blue << 0.5, 0.7, 1.0;
            EiVector3d sky = (1.0 - a) * white + a * blue;
            int spp = gen.integer(1, 65);
            EiVector3d pixel_color = EiVector3d::Zero();
            for (int k = 0; k < spp; ++k) pixel_color += (k % 2 ? c : sky);
            double gray = 0.2126 * pixel_color[0] + 0.7152 * pixel_color[1] + 0.0722 * pixel_color[2];
            int gray_byte = int(gray / spp * 255.99);
            for (int k = 0; k < 3; ++k) {
                bary_in[3 * q + k] = rec.barycentric_coordinates(k);
                fc_in[3 * q + k] = rec.face_color(k);
                hit_out[3 * q + k] = c(k);
                sky_out[3 * q + k] = sky(k);
                pc_in[3 * q + k] = pixel_color(k);
            }
            diry_in[q] = dir_y;
            spp_in[q] = spp;
            inten_out[q] = gray / spp;
            byte_out[q] = int32_t(static_cast<uint8_t>(gray_byte));
        }
        dump("shade_bary.f64", bary_in);
        dump("shade_fc.f64", fc_in);
        dump("shade_hit.f64", hit_out);
        dump("shade_diry.f64", diry_in);
        dump("shade_sky.f64", sky_out);
        dump("quant_pc.f64", pc_in);
        dump("quant_spp.i32", spp_in);
        dump("quant_intensity.f64", inten_out);
        dump("quant_byte.i32", byte_out);
    }
    return 0;
}
