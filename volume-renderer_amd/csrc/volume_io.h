// volume_io.h -- on-disk volume formats either side of the ray-march path (SURVEY 8(f)
// rows N1, N2, N4).  Host-only code, 64-bit sizes end to end.
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

namespace vr {

// ---- N1: .raw + .raw.inf sidecar (src/RendererCore.cpp:46-54,247-342)
struct RawInf {
    int dims[3] = {0, 0, 0};
    float spacing[3] = {0, 0, 0};
};
// Returns false and fills title/msg (the reference's GUI strings) on a bad sidecar.
bool parseRawInf(const std::string &inf_path, RawInf &out, std::string &title, std::string &msg);
bool writeRawInf(const std::string &inf_path, const int dims[3], const float spacing[3]);
bool readRawFile(const std::string &path, uint64_t n_bytes, std::vector<uint8_t> &out);

// ---- N2: PVM / DDS decoder (src/ddsbase.cpp:394-452,550-594,768-858), read-only
struct PvmVolume {
    std::vector<uint8_t> data;          // width*height*depth*components bytes
    unsigned width = 0, height = 0, depth = 0, components = 0;
    float scalex = 1, scaley = 1, scalez = 1;
};
bool readPVMvolume(const std::string &path, PvmVolume &out, std::string &err);
// in-memory entry points (used by tests)
bool decodeDDS(const uint8_t *chunk, size_t size, std::vector<uint8_t> &out, std::string &err);
bool parsePVM(const std::vector<uint8_t> &raw, PvmVolume &out, std::string &err);

// ---- N4: image writers for saveImage (src/RendererCore.cpp:165-182)
bool writePNG(const std::string &path, int w, int h, const uint8_t *rgb, int stride_bytes);
bool writeBMP(const std::string &path, int w, int h, const uint8_t *rgb, int stride_bytes);
bool writePPM(const std::string &path, int w, int h, const uint8_t *rgb, int stride_bytes);
// baseline JPEG, YCbCr 4:4:4, quality 1..100 (the reference writes 100, src/RendererCore.cpp:177)
bool writeJPEG(const std::string &path, int w, int h, const uint8_t *rgb, int stride_bytes, int quality);

// ---- N3: natural cubic spline transfer function (src/CubicSpline.cpp:13-115)
// knots: iso[n] ascending in 0..255, rgba[n*4]; lut: 256 x RGBA clamped to [0,1]
bool buildSplineLUT(const int32_t *iso, const float *rgba4, int n, std::vector<float> &lut256x4);

}  // namespace vr
