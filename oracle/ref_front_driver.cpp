// =============================================================================
// oracle/ref_front_driver.cpp -- TEST INFRASTRUCTURE ONLY.
//
// The reference's own front ends -- main_vsfm.cpp, main_colmap.cpp, main_bundler.cpp -- compiled in place from
// /root/reference (oracle/Makefile: _ref/libl3d_ref_front.so; their main() renamed in the object file, no source
// stored) against oracle/ref_shim_front/: a RECORDER with the public interface of L3DPP::Line3D, a stand-in for the
// part of TCLAP they use, cv::imread that only knows an image's size, and the Eigen / Boost stand-ins of ref_shim/.
// Running one of them on an SfM result leaves the sequence of calls it makes on Line3D -- constructor, undistortImage,
// addImage (camera id, K, R, t, median depth, worldpoint ids), matchImages, reconstruct3Dlines, the writers -- as JSON.
// tests/test_front_ends_pinned.py holds the library's readers of those formats (l3d_nvm_*, l3d_sfm_* in l3d_io.hip and
// their Python twins) against it: that is what pins them on the reference's code rather than on self-written files.
// =============================================================================
#include <cstring>
#include <string>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "line3D.h"   // ref_shim_front/line3D.h

// the three main() functions, renamed in their object files (objcopy --redefine-sym: C linkage, like main itself)
extern "C" int l3d_ref_main_vsfm(int argc, char** argv);
extern "C" int l3d_ref_main_colmap(int argc, char** argv);
extern "C" int l3d_ref_main_bundler(int argc, char** argv);

namespace {
std::string g_json;
struct Quiet {  // the front ends print their progress to std::cout / std::cerr
    std::streambuf *o, *e;
    std::ostringstream so, se;
    Quiet() : o(std::cout.rdbuf(so.rdbuf())), e(std::cerr.rdbuf(se.rdbuf())) {}
    ~Quiet() { std::cout.rdbuf(o); std::cerr.rdbuf(e); }
};
}  // namespace

extern "C" {

// the reference's Line3D::rotationFromQ: void f(const double q[4], double R[9]) (oracle/_ref/libl3d_ref.so: lo_ref_rotation_from_q)
void lo_front_set_rotation(void* fn) { l3d_front::log().rot = (l3d_front::rotation_fn)fn; }

// runs main_<which>.cpp's main with the given arguments (argv[0] is supplied here); returns its return value, or -100
// for an unknown name, -101 if it threw.  The calls it made are then in lo_front_log().
int lo_front_run(const char* which, int argc, const char** args) {
    l3d_front::Log& L = l3d_front::log();
    L.js.str(""); L.js.clear(); L.first = true;
    std::vector<std::string> store;
    store.push_back(std::string("main_") + which);
    for (int i = 0; i < argc; ++i) store.push_back(args[i]);
    std::vector<char*> argv;
    for (size_t i = 0; i < store.size(); ++i) argv.push_back(&store[i][0]);
    argv.push_back(0);
#ifdef _OPENMP
    omp_set_num_threads(1);   // the front ends add their images in a parallel loop: one thread = file order
#endif
    int rc = -100;
    try {
        Quiet q;
        if (!std::strcmp(which, "vsfm")) rc = l3d_ref_main_vsfm((int)store.size(), argv.data());
        else if (!std::strcmp(which, "colmap")) rc = l3d_ref_main_colmap((int)store.size(), argv.data());
        else if (!std::strcmp(which, "bundler")) rc = l3d_ref_main_bundler((int)store.size(), argv.data());
    } catch (...) { rc = -101; }
    g_json = "[" + L.js.str() + "]";
    return rc;
}

const char* lo_front_log() { return g_json.c_str(); }

}  // extern "C"
