"""Instruments a copy of rt_bvh.cpp with rdtsc accumulators around the builder's phases (centroid bounds, binning, SAH
evaluation, sides from the bins, partition; record init, tree, flatten) for profiles/tools/build_phases/run.sh.
Usage: instrument.py <out.cpp> <rt_bvh.cpp>.  The anchors are lines of the source: an assert names the one that moved."""
import sys
s=open(sys.argv[2]).read()
def rep(old,new):
    global s
    assert old in s, old
    s=s.replace(old,new,1)
rep("namespace rt {\n","#include <x86intrin.h>\nnamespace rt { unsigned long long P[16]; \n")
rep("    if (count <= 2) return false;\n    Box cb;","    if (count <= 2) return false;\n    unsigned long long T0=__rdtsc();\n    Box cb;")
rep("    int axis = 0;\n    double ext = cb.extent(0);","    unsigned long long T1=__rdtsc(); P[0]+=T1-T0;\n    int axis = 0;\n    double ext = cb.extent(0);")
rep("    // suffix (right-side) areas and counts first","    unsigned long long T2=__rdtsc(); P[1]+=T2-T1;\n    // suffix (right-side) areas and counts first")
rep("    const double width = ext / kBins;\n    pos_out = cmin + width * (best_bin + 1);","    const double width = ext / kBins;\n    pos_out = cmin + width * (best_bin + 1);\n    unsigned long long T3=__rdtsc(); P[2]+=T3-T2;")
rep("            *sides_ready = true;\n        }\n    }\n    return true;","            *sides_ready = true;\n        }\n    }\n    P[3]+=__rdtsc()-T3;\n    return true;")
rep("    int64_t mid = first, end = first + count;\n    static const bool par_partition","    unsigned long long T4=__rdtsc();\n    int64_t mid = first, end = first + count;\n    static const bool par_partition")
rep("    const int64_t nl = mid - first;\n    if (nl == 0 || nl == count) return -1;\n    return nl;","    const int64_t nl = mid - first;\n    P[4]+=__rdtsc()-T4; P[5]+=sides_ready; P[6]++;\n    if (nl == 0 || nl == count) return -1;\n    return nl;")
rep("    Elements el{nullptr, nullptr, rec.data()};\n    const double t_b1 = now_s();\n    build_tree(b.nodes, el, b.sorted_to_elem.data(), E, 4, b.midpoint_fallbacks);","    Elements el{nullptr, nullptr, rec.data()};\n    const double t_b1 = now_s();\n    P[7]+=__rdtsc()-TB;\n    unsigned long long TA=__rdtsc();\n    build_tree(b.nodes, el, b.sorted_to_elem.data(), E, 4, b.midpoint_fallbacks);\n    P[8]+=__rdtsc()-TA;")
rep("    const double t_b0 = now_s();\n    FlatVec<Rec> rec","    const double t_b0 = now_s();\n    unsigned long long TB=__rdtsc();\n    FlatVec<Rec> rec")
rep("    // ---- flatten ----\n","    // ---- flatten ----\n    unsigned long long TF=__rdtsc();\n")
rep("    sc->build_seconds = std::chrono","    P[9]+=__rdtsc()-TF;\n    sc->build_seconds = std::chrono")
open(sys.argv[1],'w').write(s)
