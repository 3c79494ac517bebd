// rtmaincpp_core.hpp -- the nanobind-free part of the `rtmaincpp` binding: flattens the ten arguments of the
// reference's render_scene (cpp/rtmain.cpp:29-38) into the pointer/size arrays of the C ABI rt_render_scene
// (include/rtgpu.h).  Templated on the container types so that it compiles -- and is tested, see
// tests/test_abi.py -- without nanobind or Eigen; bindings/rtmaincpp_nb.cpp instantiates it with
// nb::ndarray / nb::DRef<Eigen::...>.
//   NdArray : .data() -> const double*, .shape(i) -> size_t           (nb::ndarray<const double, nb::c_contig>)
//   Vec3    : operator()(i), i < 3                                     (nb::DRef<EiVector3d>)
//   Mat23   : operator()(r, c), r < 2, c < 3                           (nb::DRef<Matrix<double,2,3,RowMajor>>)
#pragma once
#include <cstdint>
#include <filesystem>
#include <stdexcept>
#include <string>
#include <vector>

#include "rtgpu.h"

namespace rtgpu_binding {

template <class NdArray, class Vec3, class Mat23>
void render_scene(const int image_height, const int image_width, const int number_of_samples,
                  const std::filesystem::path& output_directory, const int timestep_count,
                  const std::vector<NdArray>& scene_coords_expanded, const std::vector<NdArray>& scene_face_colors,
                  const std::vector<Vec3>& camera_centers, const std::vector<Vec3>& pixel_00_centers,
                  const std::vector<Mat23>& matrix_pixel_spacings, int device = 0, int rank = 0, int world = 1) {
    const size_t M = scene_coords_expanded.size(), C = camera_centers.size();
    if (scene_face_colors.size() != M) throw std::invalid_argument("cpp_render_scene: coordinate and colour lists differ in length");
    if (pixel_00_centers.size() != C || matrix_pixel_spacings.size() != C)
        throw std::invalid_argument("cpp_render_scene: camera lists differ in length");
    std::vector<const double*> coords(M ? M : 1), colors(M ? M : 1);
    std::vector<int64_t> counts(M ? M : 1);
    for (size_t m = 0; m < M; ++m) {
        coords[m] = scene_coords_expanded[m].data();
        colors[m] = scene_face_colors[m].data();
        counts[m] = (int64_t)scene_face_colors[m].shape(1);  // the reference takes E from here, rtbvh.cpp:615
        if ((int64_t)scene_face_colors[m].shape(0) < timestep_count || (int64_t)scene_coords_expanded[m].shape(0) < timestep_count)
            throw std::invalid_argument("cpp_render_scene: a mesh has fewer timesteps than timestep_count");
        if ((int64_t)scene_coords_expanded[m].shape(1) != counts[m])
            throw std::invalid_argument("cpp_render_scene: coords (T,E,3,3) and colours (T,E,3) disagree on E");
    }
    std::vector<double> cam(3 * C + 1), p00(3 * C + 1), sp(6 * C + 1);
    for (size_t c = 0; c < C; ++c) {
        for (int i = 0; i < 3; ++i) {
            cam[3 * c + i] = camera_centers[c](i);
            p00[3 * c + i] = pixel_00_centers[c](i);
        }
        for (int r = 0; r < 2; ++r)
            for (int i = 0; i < 3; ++i) sp[6 * c + 3 * r + i] = matrix_pixel_spacings[c](r, i);
    }
    const std::string dir = output_directory.string();
    const int rc = rt_render_scene(image_height, image_width, number_of_samples, dir.c_str(), timestep_count, (int)M,
                                   coords.data(), colors.data(), counts.data(), (int)C, cam.data(), p00.data(), sp.data(),
                                   device, rank, world);
    if (rc != RT_OK) throw std::runtime_error(std::string("rt_render_scene: ") + rt_last_error());
}

}  // namespace rtgpu_binding
