// main.cpp — command-line front end, same flow and flags as the reference (RestOfLife/main.cpp:29-165):
//   -s scene  -ns samples  -dx width  -dy height  -h  -v  -g
// plus what the benchmark configurations need and the reference hard-wires:
//   -d depth (reference: 20, Director.cpp:42)   -seed N   -rng philox|lcg   -gpu id   -gpus N   -o file.ppm|file.png|file.pfm
// The reference's resolution / sample clamps (main.cpp:21-27) are widened so that 200x200 and
// 7680x4320 are reachable, and its scene range bug (only scene 4 selectable, main.cpp:69) is not kept.
#include <chrono>
#include <cstdlib>
#include <iostream>
#include <stdexcept>
#include <string>

#include "Director.h"
#include "InputParser.h"

#define Nx_MIN (16)
#define Ny_MIN (16)
#define Nx_MAX (16384)
#define Ny_MAX (16384)
#define Nscene_MAX (4)
#define Ns_MAX (1024 * 1024)

namespace {
// parse an integer option; returns false (and leaves `out`) when absent or malformed
bool intOption(const InputParser& in, const char* flag, const char* what, int& out) {
    const std::string& v = in.getCmdOption(flag);
    if (v.empty()) return false;
    try {
        out = std::stoi(v, nullptr, 0);
        return true;
    } catch (const std::exception&) {
        std::cerr << "Invalid " << what << ": " << v << std::endl;
        return false;
    }
}
int clampWarn(const char* what, int x, int lo, int hi) {
    if (x >= lo && x <= hi) return x;
    int y = x < lo ? lo : hi;
    std::cerr << "WARNING: " << what << " " << x << " out of range. Using a value of " << y << std::endl;
    return y;
}
}  // namespace

int main(int argc, char* argv[]) {
    int Nx = 1200, Ny = 600, Nscene = 4, Ns = 20, depth = 20, gpu = 0, gpus = 1;  // the reference's defaults (main.cpp:32-37)
    uint32_t seed = 0x6314759u;
    int rng = RTW_RNG_PHILOX;

    InputParser cl_input(argc, argv);
    if (cl_input.cmdOptionExists("-h") || cl_input.cmdOptionExists("--help")) {
        std::cerr << "\n HELP - " << argv[0] << "\n"
                  << R"(
    -s N           Scene Selection number N (0 Cornell box, 1 moving spheres, 2 spheres with a light,
                   3 Cornell box with volumes, 4 The Next Week final scene)
    -ns N          Sample each pixel N times (N: 1, 2, etc.)
    -dx Nx         Output image width (x dimension)
    -dy Ny         Output image height (y dimension)
    -d N           Maximum path depth (reference: 20)
    -seed N        RNG seed
    -rng K         philox (default) or lcg (the reference's tea+lcg generator)
    -denoise N     N passes (1..8) of the a-trous filter that stands in for the reference's AI denoiser (default: off)
    -est K         reference (default: the reference's estimator, quirks included), corrected, brute
                   (corrected without light sampling) or mixture (the book's 50/50 mixture of light and cosine sampling)
    -cam K         perspective (default), environment or orthographic (the reference's two unused camera kinds, scene/ioCamera.h:118-179)
    -gpu N         Device ordinal (the first one with -gpus)
    -gpus N        Render on N GPUs of this node: interleaved row shards, one gather onto the first device (default 1)
    -o FILE        Write FILE instead of ASCII P3 on stdout: *.ppm = binary P6, *.png = 8-bit PNG, *.pfm = linear float PFM

    -h             This help message.
    -v             Verbose output.
    -g             Debug output.

)";
        return EXIT_SUCCESS;
    }
    const bool Qverbose = cl_input.cmdOptionExists("-v");
    const bool Qdebug = cl_input.cmdOptionExists("-g");

    int x;
    if (intOption(cl_input, "-s", "scene number", x)) {
        if (x >= 0 && x <= Nscene_MAX) Nscene = x;
        else {
            std::cerr << "WARNING: Scene number " << x << " out of range. Maximum scene number: " << Nscene_MAX << std::endl;
            std::cerr << "WARNING: Using a scene value of " << Nscene << std::endl;
        }
    }
    if (intOption(cl_input, "-ns", "number of samples", x)) Ns = clampWarn("Number of samples", x, 1, Ns_MAX);
    if (intOption(cl_input, "-dx", "image width (-dx)", x)) Nx = clampWarn("Width (-dx)", x, Nx_MIN, Nx_MAX);
    if (intOption(cl_input, "-dy", "image height (-dy)", x)) Ny = clampWarn("Height (-dy)", x, Ny_MIN, Ny_MAX);
    if (intOption(cl_input, "-d", "depth (-d)", x)) depth = clampWarn("Depth (-d)", x, 1, 1 << 20);
    if (intOption(cl_input, "-gpu", "device (-gpu)", x)) gpu = x;
    if (intOption(cl_input, "-gpus", "number of devices (-gpus)", x)) gpus = clampWarn("Number of devices (-gpus)", x, 1, 64);
    if (intOption(cl_input, "-seed", "seed", x)) seed = static_cast<uint32_t>(x);
    const std::string& rngName = cl_input.getCmdOption("-rng");
    if (rngName == "lcg") rng = RTW_RNG_TEA_LCG;
    else if (!rngName.empty() && rngName != "philox") std::cerr << "WARNING: unknown -rng " << rngName << ", using philox" << std::endl;

    int estimator = RTW_EST_REFERENCE;
    const std::string& estName = cl_input.getCmdOption("-est");
    if (estName == "corrected") estimator = RTW_EST_CORRECTED;
    else if (estName == "brute") estimator = RTW_EST_CORRECTED_NO_NEE;
    else if (estName == "mixture") estimator = RTW_EST_MIXTURE;
    else if (!estName.empty() && estName != "reference") std::cerr << "WARNING: unknown -est " << estName << ", using reference" << std::endl;

    int camKind = RTW_CAM_PERSPECTIVE;
    const std::string& camName = cl_input.getCmdOption("-cam");
    if (camName == "environment") camKind = RTW_CAM_ENVIRONMENT;
    else if (camName == "orthographic") camKind = RTW_CAM_ORTHOGRAPHIC;
    else if (!camName.empty() && camName != "perspective") std::cerr << "WARNING: unknown -cam " << camName << ", using perspective" << std::endl;

    Director director(Qverbose, Qdebug);
    director.setCameraKind(camKind);
    // RTW_SAME_DEVICE=1 (tests on a one-GPU box): all -gpus N shards render on device -gpu
    if (gpus > 1) director.setDevices(gpus, gpu, std::getenv("RTW_SAME_DEVICE") != nullptr);
    else director.setDevice(gpu);
    director.setMaxDepth(depth);
    director.setSeed(seed);
    director.setRngKind(rng);
    director.setEstimator(estimator);
    if (intOption(cl_input, "-denoise", "denoise passes", x)) director.setDenoise(clampWarn("Denoise passes", x, 0, 8), 0.5f);

    auto start = std::chrono::system_clock::now();
    director.init(Nx, Ny, Ns);
    if (Qverbose) {
        std::cerr << "INFO: Output image dimensions: " << Nx << 'x' << Ny << std::endl;
        std::cerr << "INFO: Number of rays sent per pixel: " << Ns << std::endl;
        std::cerr << "INFO: Scene number selected: " << Nscene << std::endl;
    }
    director.createScene(Nscene);
    director.renderFrame();
    auto stop = std::chrono::system_clock::now();
    std::cerr << "INFO: Took " << std::chrono::duration<float>(stop - start).count() << " seconds." << std::endl;

    const std::string& outPath = cl_input.getCmdOption("-o");
    if (outPath.empty()) {
        director.printPPM();
    } else {
        auto ends = [&](const char* ext) { return outPath.size() > 4 && outPath.compare(outPath.size() - 4, 4, ext) == 0; };
        const bool ok = ends(".pfm") ? director.writePFM(outPath) : ends(".png") ? director.writePNG(outPath) : director.writeBinaryPPM(outPath);
        if (!ok) {
            std::cerr << "ERROR: cannot write " << outPath << std::endl;
            director.destroy();
            return EXIT_FAILURE;
        }
    }
    director.destroy();
    return EXIT_SUCCESS;
}
