// Test program for bindings/rtmaincpp_core.hpp: instantiates the binding's nanobind-free core with plain C++
// containers shaped like nb::ndarray / Eigen row vectors, and calls it.
// usage: probe <out_dir> <T> [cam(3) p00(3) spacing(6) as C99 hex floats]
// exit 0 = rendered; exit 3 = rt_render_scene reported an error (message on stdout).
#include <array>
#include <cstdio>
#include <cmath>
#include <cstdlib>

#include "rtmaincpp_core.hpp"

struct Nd {  // like nb::ndarray<const double, nb::c_contig>
    std::vector<double> v;
    std::vector<size_t> shp;
    const double* data() const { return v.data(); }
    size_t shape(size_t i) const { return shp[i]; }
};
struct V3 {
    std::array<double, 3> a;
    double operator()(int i) const { return a[(size_t)i]; }
};
struct M23 {
    std::array<double, 6> a;
    double operator()(int r, int c) const { return a[(size_t)(3 * r + c)]; }
};

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    const int T = std::atoi(argv[2]);
    // one quad (2 triangles) at z = -2 facing +z, moving in x with the timestep; colours 0.25 / 0.75
    Nd coords, colors;
    coords.shp = {(size_t)T, 2, 3, 3};
    colors.shp = {(size_t)T, 2, 3};
    for (int t = 0; t < T; ++t) {
        const double dx = 0.1 * t;
        const double q[4][3] = {{-1 + dx, -1, -2}, {1 + dx, -1, -2}, {1 + dx, 1, -2}, {-1 + dx, 1, -2}};
        const int idx[2][3] = {{0, 1, 2}, {0, 2, 3}};
        for (auto& tri : idx)
            for (int k : tri)
                for (int c = 0; c < 3; ++c) coords.v.push_back(q[k][c]);
        for (int e = 0; e < 2; ++e)
            for (int c = 0; c < 3; ++c) colors.v.push_back(e ? 0.75 : 0.25);
    }
    // camera of a 16 x 8 image: the caller's 12 values (bit-exact hex floats), else fov 90 at the origin looking down -z
    const int W = 16, H = 8;
    const double vh = 2.0, vw = vh * W / H;
    std::vector<V3> cam{{{0, 0, 0}}}, p00{{{-vw / 2 + 0.5 * vw / W, vh / 2 - 0.5 * vh / H, -1.0}}};
    std::vector<M23> sp{{{vw / W, 0, 0, 0, -vh / H, 0}}};
    if (argc >= 15) {
        for (int i = 0; i < 3; ++i) cam[0].a[(size_t)i] = std::strtod(argv[3 + i], nullptr);
        for (int i = 0; i < 3; ++i) p00[0].a[(size_t)i] = std::strtod(argv[6 + i], nullptr);
        for (int i = 0; i < 6; ++i) sp[0].a[(size_t)i] = std::strtod(argv[9 + i], nullptr);
    }
    try {
        rtgpu_binding::render_scene(H, W, 2, std::filesystem::path(argv[1]), T, std::vector<Nd>{coords},
                                    std::vector<Nd>{colors}, cam, p00, sp);
    } catch (const std::exception& e) {
        std::printf("error: %s\n", e.what());
        return 3;
    }
    return 0;
}
