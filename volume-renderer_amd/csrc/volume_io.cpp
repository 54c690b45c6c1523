// volume_io.cpp -- see volume_io.h.  Clean-room implementations written from the
// format behaviour of the reference (file:line cited per function); the reference's
// PVM/DDS code is GPL and is not reproduced.
#include "volume_io.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>

namespace vr {

// =====================================================================  N1: RAW
// Sidecar grammar (src/RendererCore.cpp:256-289): blank lines skipped; a line equal
// to "#dimensions" is followed by "X Y Z"; "#voxel-spacing" by "sx sy sz"; anything
// else is ignored.
bool parseRawInf(const std::string &inf_path, RawInf &out, std::string &title, std::string &msg)
{
    std::ifstream in(inf_path);
    if (!in) {
        title = "Error!";
        msg = "Failed to open .raw.inf file.";
        return false;
    }
    out = RawInf();
    std::string line;
    while (std::getline(in, line)) {
        if (line.empty()) continue;
        const bool dims = (line == "#dimensions"), spacing = (line == "#voxel-spacing");
        if (!dims && !spacing) continue;
        std::string values;
        std::getline(in, values);
        if (values.empty()) {
            title = "Invalid .raw.inf file!";
            msg = dims ? "Dimensions for Volume Data not provided in \"raw.inf\" file."
                       : "Aspect Ratio for Volume Data not provided in \"raw.inf\" file.";
            return false;
        }
        std::stringstream ss(values);
        if (dims) ss >> out.dims[0] >> out.dims[1] >> out.dims[2];
        else ss >> out.spacing[0] >> out.spacing[1] >> out.spacing[2];
    }
    if (out.dims[0] == 0 && out.dims[1] == 0 && out.dims[2] == 0) {
        title = "Invalid .raw.inf file!";
        msg = "Dimensions for Volume Data not provided in \"raw.inf\" file. Make sure the header is "
              "\"#dimesnsions\"";
        return false;
    }
    if (out.spacing[0] == 0 && out.spacing[1] == 0 && out.spacing[2] == 0) {
        title = "Invalid .raw.inf file!";
        msg = "Aspect Ratio for Volume Data not provided in \"raw.inf\" file. Make sure the header is "
              "\"#voxel-spacing\"";
        return false;
    }
    return true;
}

// src/RendererCore.cpp:304-317
bool writeRawInf(const std::string &inf_path, const int dims[3], const float spacing[3])
{
    std::ofstream out(inf_path);
    if (!out) return false;
    out << "#dimensions\n" << dims[0] << " " << dims[1] << " " << dims[2] << "\n\n"
        << "#voxel-spacing\n" << spacing[0] << " " << spacing[1] << " " << spacing[2] << std::endl;
    return bool(out);
}

// src/RendererCore.cpp:318-341.  Like the reference a short file leaves the tail zero.
bool readRawFile(const std::string &path, uint64_t n_bytes, std::vector<uint8_t> &out)
{
    std::ifstream in(path, std::ios::binary);
    if (!in) return false;
    out.assign(n_bytes, 0);
    in.read(reinterpret_cast<char *>(out.data()), (std::streamsize)n_bytes);
    return true;
}

// =====================================================================  N2: PVM
namespace {

// MSB-first bit reader over big-endian 32-bit words; reads past the end give zeros
// (src/ddsbase.cpp:118-158).
class BitReader {
public:
    BitReader(const uint8_t *p, size_t n) : p_(p), n_(n) {}
    uint32_t read(unsigned bits)
    {
        uint32_t v = 0;
        while (bits > 0) {
            if (avail_ == 0) refill();
            const unsigned take = bits < avail_ ? bits : avail_;
            const uint32_t chunk = (take == 32) ? word_ : ((word_ >> (avail_ - take)) & ((1u << take) - 1u));
            v = (take == 32) ? chunk : ((v << take) | chunk);
            avail_ -= take;
            bits -= take;
        }
        return v;
    }

private:
    void refill()
    {
        word_ = 0;
        for (int i = 0; i < 4; i++) {
            word_ <<= 8;
            if (pos_ < n_) word_ |= p_[pos_];
            pos_++;
        }
        avail_ = 32;
    }
    const uint8_t *p_;
    size_t n_, pos_ = 0;
    uint32_t word_ = 0;
    unsigned avail_ = 0;
};

// Undo the byte de-interleave of the encoder (src/ddsbase.cpp:167-229 with
// restore=TRUE): within each block the stream stores lane 0 of every `skip`-byte
// group first, then lane 1, ...
void restoreInterleave(std::vector<uint8_t> &data, unsigned skip, uint64_t block)
{
    if (skip <= 1 || data.empty()) return;
    const uint64_t bytes = data.size();
    const uint64_t chunk = block == 0 ? bytes : (uint64_t)skip * block;
    std::vector<uint8_t> tmp;
    for (uint64_t base = 0; base < bytes; base += chunk) {
        const uint64_t len = (bytes - base < chunk) ? bytes - base : chunk;
        tmp.resize(len);
        uint64_t src = base;
        for (unsigned lane = 0; lane < skip; lane++)
            for (uint64_t j = lane; j < len; j += skip) tmp[j] = data[src++];
        std::memcpy(data.data() + base, tmp.data(), len);
    }
}

const char kMagicV3d[] = "DDS v3d\n";
const char kMagicV3e[] = "DDS v3e\n";

}  // namespace

// src/ddsbase.cpp:394-452 (DDS_decode) + :550-594 (readDDSfile magic / block choice)
bool decodeDDS(const uint8_t *file, size_t size, std::vector<uint8_t> &out, std::string &err)
{
    uint64_t block;
    if (size >= 8 && std::memcmp(file, kMagicV3d, 8) == 0) block = 0;
    else if (size >= 8 && std::memcmp(file, kMagicV3e, 8) == 0) block = 1u << 24;   // DDS_INTERLEAVE
    else { err = "not a DDS stream"; return false; }
    BitReader br(file + 8, size - 8);
    const unsigned skip = br.read(2) + 1;
    const uint64_t strip = br.read(16) + 1;
    out.clear();
    int act = 0;
    for (;;) {
        const unsigned run = br.read(7);                 // DDS_RL
        if (run == 0) break;
        unsigned bits = br.read(3);
        if (bits >= 1) bits += 1;                        // DDS_decode(bits)
        const int bias = (1 << bits) / 2;
        for (unsigned r = 0; r < run; r++) {
            const uint64_t cnt = out.size();
            const int delta = (int)br.read(bits) - bias;
            if (strip == 1 || cnt <= strip) act += delta;
            else act += (int)out[cnt - strip] - (int)out[cnt - strip - 1] + delta;
            act &= 255;                                  // wrap into 0..255
            out.push_back((uint8_t)act);
        }
    }
    restoreInterleave(out, skip, block);
    return true;
}

// src/ddsbase.cpp:768-858 (header grammar PVM / PVM2 / PVM3)
bool parsePVM(const std::vector<uint8_t> &raw, PvmVolume &out, std::string &err)
{
    if (raw.size() < 5) { err = "PVM too short"; return false; }
    std::string text(reinterpret_cast<const char *>(raw.data()), raw.size());   // may hold NULs
    text.push_back('\0');
    const char *base = text.c_str();
    const char *ptr;
    int version = 1;
    int w = 0, h = 0, d = 0, numc = 0;
    float sx = 1.0f, sy = 1.0f, sz = 1.0f;
    auto next_line = [&](const char *p) -> const char * {
        const char *nl = std::strchr(p, '\n');
        return nl ? nl + 1 : nullptr;
    };
    if (std::strncmp(base, "PVM\n", 4) == 0) {
        ptr = base + 4;
        while (*ptr == '#') {
            ptr = next_line(ptr);
            if (!ptr) { err = "PVM: unterminated comment"; return false; }
        }
        if (std::sscanf(ptr, "%d %d %d\n", &w, &h, &d) != 3) { err = "PVM: bad dimensions"; return false; }
    } else {
        if (std::strncmp(base, "PVM2\n", 5) == 0) version = 2;
        else if (std::strncmp(base, "PVM3\n", 5) == 0) version = 3;
        else { err = "not a PVM volume"; return false; }
        ptr = base + 5;
        if (std::sscanf(ptr, "%d %d %d\n%g %g %g\n", &w, &h, &d, &sx, &sy, &sz) != 6) {
            err = "PVM: bad dimensions/scale"; return false;
        }
        if (sx <= 0.0f || sy <= 0.0f || sz <= 0.0f) { err = "PVM: non-positive scale"; return false; }
        ptr = next_line(ptr);
        if (!ptr) { err = "PVM: truncated header"; return false; }
    }
    if (w < 1 || h < 1 || d < 1) { err = "PVM: non-positive dimensions"; return false; }
    ptr = next_line(ptr);
    if (!ptr || std::sscanf(ptr, "%d\n", &numc) != 1 || numc < 1) { err = "PVM: bad component count"; return false; }
    ptr = next_line(ptr);
    if (!ptr) { err = "PVM: truncated header"; return false; }
    const uint64_t payload = (uint64_t)w * (uint64_t)h * (uint64_t)d * (uint64_t)numc;
    const uint64_t offset = (uint64_t)(ptr - base);
    if (offset + payload > raw.size()) { err = "PVM: payload shorter than header says"; return false; }
    uint64_t tail = 0;
    if (version == 3) {
        // four NUL-terminated strings: description, courtesy, parameter, comment
        uint64_t p = offset + payload;
        for (int s = 0; s < 4; s++) {
            while (p < raw.size() && raw[p] != 0) p++;
            p++;   // the terminator
        }
        tail = p - (offset + payload);
    }
    if (offset + payload + tail != raw.size()) { err = "PVM: size mismatch"; return false; }
    out.width = (unsigned)w; out.height = (unsigned)h; out.depth = (unsigned)d; out.components = (unsigned)numc;
    out.scalex = sx; out.scaley = sy; out.scalez = sz;
    out.data.assign(raw.begin() + (std::ptrdiff_t)offset, raw.begin() + (std::ptrdiff_t)(offset + payload));
    return true;
}

bool readPVMvolume(const std::string &path, PvmVolume &out, std::string &err)
{
    std::ifstream in(path, std::ios::binary);
    if (!in) { err = "cannot open " + path; return false; }
    std::vector<uint8_t> file((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    std::vector<uint8_t> raw;
    std::string dds_err;
    if (!decodeDDS(file.data(), file.size(), raw, dds_err)) raw.swap(file);   // plain (uncompressed) PVM
    return parsePVM(raw, out, err);
}

// =====================================================================  N4: images
namespace {
uint32_t crc32_update(uint32_t crc, const uint8_t *p, size_t n)
{
    static uint32_t table[256];
    static bool init = false;
    if (!init) {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t c = i;
            for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
        init = true;
    }
    for (size_t i = 0; i < n; i++) crc = table[(crc ^ p[i]) & 0xff] ^ (crc >> 8);
    return crc;
}
void put_be32(std::vector<uint8_t> &v, uint32_t x)
{
    v.push_back((uint8_t)(x >> 24)); v.push_back((uint8_t)(x >> 16)); v.push_back((uint8_t)(x >> 8)); v.push_back((uint8_t)x);
}
void png_chunk(std::vector<uint8_t> &out, const char type[4], const std::vector<uint8_t> &body)
{
    put_be32(out, (uint32_t)body.size());
    const size_t start = out.size();
    out.insert(out.end(), type, type + 4);
    out.insert(out.end(), body.begin(), body.end());
    const uint32_t crc = crc32_update(0xffffffffu, out.data() + start, out.size() - start) ^ 0xffffffffu;
    put_be32(out, crc);
}
}  // namespace

// 8-bit RGB PNG with stored (uncompressed) deflate blocks; rows are written in the
// order given (the caller flips, like stbi_flip_vertically_on_write(1)).
bool writePNG(const std::string &path, int w, int h, const uint8_t *rgb, int stride)
{
    std::vector<uint8_t> raw;
    raw.reserve((size_t)h * ((size_t)w * 3 + 1));
    for (int y = 0; y < h; y++) {
        raw.push_back(0);   // filter: none
        raw.insert(raw.end(), rgb + (size_t)y * stride, rgb + (size_t)y * stride + (size_t)w * 3);
    }
    std::vector<uint8_t> z;
    z.push_back(0x78); z.push_back(0x01);
    size_t pos = 0;
    uint32_t a = 1, b = 0;
    for (uint8_t c : raw) { a = (a + c) % 65521u; b = (b + a) % 65521u; }
    do {
        const size_t n = raw.size() - pos < 65535 ? raw.size() - pos : 65535;
        z.push_back(pos + n == raw.size() ? 1 : 0);
        z.push_back((uint8_t)(n & 0xff)); z.push_back((uint8_t)(n >> 8));
        z.push_back((uint8_t)(~n & 0xff)); z.push_back((uint8_t)((~n >> 8) & 0xff));
        z.insert(z.end(), raw.begin() + (std::ptrdiff_t)pos, raw.begin() + (std::ptrdiff_t)(pos + n));
        pos += n;
    } while (pos < raw.size());
    put_be32(z, (b << 16) | a);
    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    std::vector<uint8_t> ihdr;
    put_be32(ihdr, (uint32_t)w); put_be32(ihdr, (uint32_t)h);
    ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    png_chunk(out, "IHDR", ihdr);
    png_chunk(out, "IDAT", z);
    png_chunk(out, "IEND", {});
    std::ofstream f(path, std::ios::binary);
    if (!f) return false;
    f.write(reinterpret_cast<const char *>(out.data()), (std::streamsize)out.size());
    return bool(f);
}

bool writeBMP(const std::string &path, int w, int h, const uint8_t *rgb, int stride)
{
    const int row = (w * 3 + 3) & ~3;
    const uint32_t size = 54 + (uint32_t)row * (uint32_t)h;
    std::vector<uint8_t> out(size, 0);
    auto le32 = [&](size_t o, uint32_t v) { out[o] = (uint8_t)v; out[o + 1] = (uint8_t)(v >> 8); out[o + 2] = (uint8_t)(v >> 16); out[o + 3] = (uint8_t)(v >> 24); };
    out[0] = 'B'; out[1] = 'M';
    le32(2, size); le32(10, 54); le32(14, 40); le32(18, (uint32_t)w); le32(22, (uint32_t)h);
    out[26] = 1; out[28] = 24; le32(34, (uint32_t)row * (uint32_t)h);
    for (int y = 0; y < h; y++) {          // BMP is bottom-up: input row 0 = top
        const uint8_t *src = rgb + (size_t)y * stride;
        uint8_t *dst = out.data() + 54 + (size_t)(h - 1 - y) * row;
        for (int x = 0; x < w; x++) { dst[3 * x] = src[3 * x + 2]; dst[3 * x + 1] = src[3 * x + 1]; dst[3 * x + 2] = src[3 * x]; }
    }
    std::ofstream f(path, std::ios::binary);
    if (!f) return false;
    f.write(reinterpret_cast<const char *>(out.data()), (std::streamsize)out.size());
    return bool(f);
}

bool writePPM(const std::string &path, int w, int h, const uint8_t *rgb, int stride)
{
    std::ofstream f(path, std::ios::binary);
    if (!f) return false;
    f << "P6\n" << w << " " << h << "\n255\n";
    for (int y = 0; y < h; y++) f.write(reinterpret_cast<const char *>(rgb + (size_t)y * stride), (std::streamsize)w * 3);
    return bool(f);
}

// =====================================================================  N3: spline
// Natural cubic spline through (iso, rgba) knots: tridiagonal forward elimination /
// back substitution per channel (src/CubicSpline.cpp:50-115), evaluated at integer
// iso values (src/CubicSpline.cpp:20-40), clamped to [0,1] as the widget draws it
// (src/UI/elements/AlphaControlSplineWidget.cpp:247).
bool buildSplineLUT(const int32_t *iso, const float *rgba4, int n, std::vector<float> &lut)
{
    if (n < 2) return false;
    for (int i = 1; i < n; i++)
        if (iso[i] <= iso[i - 1]) return false;
    const int segs = n - 1;
    lut.assign(256 * 4, 0.0f);
    std::vector<float> gamma(n), delta(n), D(n);
    for (int ch = 0; ch < 4; ch++) {
        auto y = [&](int i) { return rgba4[i * 4 + ch]; };
        gamma[0] = 0.5f;
        for (int i = 1; i < segs; i++) gamma[i] = 1.0f / ((4.0f * 1.0f) - gamma[i - 1]);
        gamma[segs] = 1.0f / ((2.0f * 1.0f) - gamma[segs - 1]);
        delta[0] = 3.0f * (y(1) - y(0)) * gamma[0];
        for (int i = 1; i < segs; i++) delta[i] = (3.0f * (y(i + 1) - y(i - 1)) - delta[i - 1]) * gamma[i];
        delta[segs] = (3.0f * (y(segs) - y(segs - 1)) - delta[segs - 1]) * gamma[segs];
        D[segs] = delta[segs];
        for (int i = segs - 1; i >= 0; i--) D[i] = delta[i] - gamma[i] * D[i + 1];
        for (int e = 0; e < 256; e++) {
            float val = 0.0f;
            bool hit = false;
            int seg = 0;
            float t = 0.0f;
            for (int i = 0; i < n; i++) {
                if (iso[i] == e) { val = y(i); hit = true; break; }
                if (iso[i] > e) {
                    seg = i - 1;
                    if (seg < 0) seg = 0;
                    t = (float)(e - iso[seg]) / (float)(iso[seg + 1] - iso[seg]);
                    break;
                }
            }
            if (!hit) {
                const float a = y(seg), b = D[seg];
                const float c = 3.0f * (y(seg + 1) - y(seg)) - 2.0f * D[seg] - D[seg + 1];
                const float d = 2.0f * (y(seg) - y(seg + 1)) + D[seg] + D[seg + 1];
                val = a + t * (b + t * (c + t * d));
            }
            lut[e * 4 + ch] = std::fmin(std::fmax(val, 0.0f), 1.0f);
        }
    }
    return true;
}

}  // namespace vr
