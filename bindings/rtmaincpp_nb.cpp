// rtmaincpp_nb.cpp -- the nanobind module `rtmaincpp` a maintainer of the reference would build in place of the
// export that is commented out at cpp/rtmain.cpp:157-160: the same name, the same ten arguments
// (cpp/rtmain.cpp:29-38), rendering through librtgpu.so.
//
// NOT BUILT IN THIS REPOSITORY'S IMAGE: nanobind is not installed here (and the reference's CMake wants
// Python 3.11 EXACT, cpp/CMakeLists.txt:17).  Everything that does not need nanobind lives in
// rtmaincpp_core.hpp, which IS compiled, linked against librtgpu.so and run by tests/test_abi.py with plain
// C++ container types.  This file is the remaining glue; build it with bindings/CMakeLists.txt where
// nanobind exists.
#include <nanobind/eigen/dense.h>
#include <nanobind/nanobind.h>
#include <nanobind/ndarray.h>
#include <nanobind/stl/filesystem.h>
#include <nanobind/stl/vector.h>

#include <Eigen/Dense>

#include "rtmaincpp_core.hpp"

namespace nb = nanobind;
using EiVector3d = Eigen::Matrix<double, 1, 3, Eigen::RowMajor>;  // cpp/rteigentypes.h:7-11
using EiMatrix23d = Eigen::Matrix<double, 2, 3, Eigen::RowMajor>;
using NdArray = nb::ndarray<const double, nb::c_contig>;

static void render_scene(const int image_height, const int image_width, const int number_of_samples,
                         const std::filesystem::path output_directory, const int timestep_count,
                         const std::vector<NdArray>& scene_coords_expanded, const std::vector<NdArray>& scene_face_colors,
                         const std::vector<nb::DRef<EiVector3d>> camera_centers,
                         const std::vector<nb::DRef<EiVector3d>> pixel_00_centers,
                         const std::vector<nb::DRef<EiMatrix23d>> matrix_pixel_spacings) {
    // the GIL can go: the C ABI touches no Python object and only reads the borrowed buffers
    nb::gil_scoped_release release;
    rtgpu_binding::render_scene(image_height, image_width, number_of_samples, output_directory, timestep_count,
                                scene_coords_expanded, scene_face_colors, camera_centers, pixel_00_centers,
                                matrix_pixel_spacings);
}

NB_MODULE(rtmaincpp, a) { a.def("cpp_render_scene", &render_scene); }
