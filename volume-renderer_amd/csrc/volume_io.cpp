// volume_io.cpp -- see volume_io.h.  Clean-room implementations written from the
// format behaviour of the reference (file:line cited per function); the reference's
// PVM/DDS code is GPL and is not reproduced.
#include "volume_io.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <initializer_list>
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
    // w*h*d*numc from four untrusted values can wrap 64 bits: bound the running product by the
    // bytes that are actually there (division, not multiplication)
    uint64_t payload = 1;
    for (const uint64_t f : {(uint64_t)w, (uint64_t)h, (uint64_t)d, (uint64_t)numc}) {
        if (f > (uint64_t)raw.size() / payload) { err = "PVM: payload shorter than header says"; return false; }
        payload *= f;
    }
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

// Baseline sequential JPEG (ITU-T T.81), 8-bit YCbCr 4:4:4, the Annex K example quantisation
// and Huffman tables, IJG-style quality scaling -- what saveImage(".jpg") needs
// (src/RendererCore.cpp:176-177 writes quality 100).  Written from the standard; rows are
// taken in the order given.
namespace {
const uint8_t kZigzag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22,
                             15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
const uint8_t kQuantLuma[64] = {16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57,
                                69, 56, 14, 17, 22, 29, 51, 87, 80, 62, 18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55,
                                64, 81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
const uint8_t kQuantChroma[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99,
                                  99, 99, 47, 66, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
                                  99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};
const uint8_t kDcLumaBits[16] = {0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
const uint8_t kDcChromaBits[16] = {0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
const uint8_t kDcVals[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
const uint8_t kAcLumaBits[16] = {0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d};
const uint8_t kAcLumaVals[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71,
    0x14, 0x32, 0x81, 0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72,
    0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37,
    0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59,
    0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83,
    0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3,
    0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3,
    0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2,
    0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};
const uint8_t kAcChromaBits[16] = {0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77};
const uint8_t kAcChromaVals[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22,
    0x32, 0x81, 0x08, 0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1,
    0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36,
    0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58,
    0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a,
    0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a,
    0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba,
    0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda,
    0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};

struct HuffCodes { uint16_t code[256]; uint8_t len[256]; };

// canonical code assignment of T.81 Annex C
HuffCodes buildHuff(const uint8_t bits[16], const uint8_t *vals)
{
    HuffCodes h = {};
    uint16_t code = 0;
    int k = 0;
    for (int l = 1; l <= 16; l++) {
        for (int i = 0; i < bits[l - 1]; i++, k++) { h.code[vals[k]] = code++; h.len[vals[k]] = (uint8_t)l; }
        code <<= 1;
    }
    return h;
}

struct JpegBits {
    std::vector<uint8_t> &out;
    uint32_t acc = 0;
    int n = 0;
    void put(uint32_t v, int len)
    {
        acc = (acc << len) | (v & ((1u << len) - 1u));
        n += len;
        while (n >= 8) {
            const uint8_t b = (uint8_t)(acc >> (n - 8));
            out.push_back(b);
            if (b == 0xff) out.push_back(0);      // byte stuffing
            n -= 8;
        }
    }
    void flush() { if (n > 0) put(0x7f, 8 - n); }   // pad with 1-bits
};

void jpegSegment(std::vector<uint8_t> &out, uint8_t marker, const std::vector<uint8_t> &body)
{
    out.push_back(0xff); out.push_back(marker);
    const size_t len = body.size() + 2;
    out.push_back((uint8_t)(len >> 8)); out.push_back((uint8_t)len);
    out.insert(out.end(), body.begin(), body.end());
}

// separable 8x8 forward DCT-II with the JPEG normalisation (double precision)
void fdct8x8(const double in[64], double outc[64])
{
    static double basis[8][8];
    static bool init = false;
    if (!init) {
        for (int u = 0; u < 8; u++)
            for (int x = 0; x < 8; x++)
                basis[u][x] = (u == 0 ? std::sqrt(0.125) : 0.5) * std::cos((2 * x + 1) * u * 3.14159265358979323846 / 16.0);
        init = true;
    }
    double tmp[64];
    for (int y = 0; y < 8; y++)
        for (int u = 0; u < 8; u++) {
            double a = 0.0;
            for (int x = 0; x < 8; x++) a += basis[u][x] * in[y * 8 + x];
            tmp[y * 8 + u] = a;
        }
    for (int v = 0; v < 8; v++)
        for (int u = 0; u < 8; u++) {
            double a = 0.0;
            for (int y = 0; y < 8; y++) a += basis[v][y] * tmp[y * 8 + u];
            outc[v * 8 + u] = a;
        }
}

int bitLength(int v) { int n = 0; for (v = v < 0 ? -v : v; v; v >>= 1) n++; return n; }
}  // namespace

bool writeJPEG(const std::string &path, int w, int h, const uint8_t *rgb, int stride, int quality)
{
    if (w <= 0 || h <= 0 || w > 65535 || h > 65535) return false;
    quality = quality < 1 ? 1 : (quality > 100 ? 100 : quality);
    const int scale = quality < 50 ? 5000 / quality : 200 - 2 * quality;
    uint8_t qt[2][64];
    for (int i = 0; i < 64; i++) {
        const int l = (kQuantLuma[i] * scale + 50) / 100, c = (kQuantChroma[i] * scale + 50) / 100;
        qt[0][i] = (uint8_t)(l < 1 ? 1 : (l > 255 ? 255 : l));
        qt[1][i] = (uint8_t)(c < 1 ? 1 : (c > 255 ? 255 : c));
    }
    std::vector<uint8_t> out = {0xff, 0xd8};
    jpegSegment(out, 0xe0, {'J', 'F', 'I', 'F', 0, 1, 1, 0, 0, 1, 0, 1, 0, 0});
    for (int t = 0; t < 2; t++) {
        std::vector<uint8_t> b = {(uint8_t)t};
        for (int i = 0; i < 64; i++) b.push_back(qt[t][kZigzag[i]]);
        jpegSegment(out, 0xdb, b);
    }
    jpegSegment(out, 0xc0, {8, (uint8_t)(h >> 8), (uint8_t)h, (uint8_t)(w >> 8), (uint8_t)w, 3,
                            1, 0x11, 0, 2, 0x11, 1, 3, 0x11, 1});
    auto dht = [&](uint8_t id, const uint8_t bits[16], const uint8_t *vals, int nvals) {
        std::vector<uint8_t> b = {id};
        b.insert(b.end(), bits, bits + 16);
        b.insert(b.end(), vals, vals + nvals);
        jpegSegment(out, 0xc4, b);
    };
    dht(0x00, kDcLumaBits, kDcVals, 12); dht(0x10, kAcLumaBits, kAcLumaVals, 162);
    dht(0x01, kDcChromaBits, kDcVals, 12); dht(0x11, kAcChromaBits, kAcChromaVals, 162);
    jpegSegment(out, 0xda, {3, 1, 0x00, 2, 0x11, 3, 0x11, 0, 63, 0});

    const HuffCodes dc[2] = {buildHuff(kDcLumaBits, kDcVals), buildHuff(kDcChromaBits, kDcVals)};
    const HuffCodes ac[2] = {buildHuff(kAcLumaBits, kAcLumaVals), buildHuff(kAcChromaBits, kAcChromaVals)};
    JpegBits bw{out};
    int pred[3] = {0, 0, 0};
    for (int by = 0; by < h; by += 8)
        for (int bx = 0; bx < w; bx += 8) {
            double comp[3][64];
            for (int y = 0; y < 8; y++)
                for (int x = 0; x < 8; x++) {
                    const int sx = bx + x < w ? bx + x : w - 1, sy = by + y < h ? by + y : h - 1;   // edge replication
                    const uint8_t *p = rgb + (size_t)sy * stride + (size_t)sx * 3;
                    const double r = p[0], g = p[1], b = p[2];
                    comp[0][y * 8 + x] = 0.299 * r + 0.587 * g + 0.114 * b - 128.0;
                    comp[1][y * 8 + x] = -0.168735892 * r - 0.331264108 * g + 0.5 * b;
                    comp[2][y * 8 + x] = 0.5 * r - 0.418687589 * g - 0.081312411 * b;
                }
            for (int c = 0; c < 3; c++) {
                const int t = c == 0 ? 0 : 1;
                double coef[64];
                fdct8x8(comp[c], coef);
                int q[64];
                for (int i = 0; i < 64; i++) q[i] = (int)std::lround(coef[kZigzag[i]] / (double)qt[t][kZigzag[i]]);
                // DC difference
                const int diff = q[0] - pred[c];
                pred[c] = q[0];
                int nb = bitLength(diff);
                bw.put(dc[t].code[nb], dc[t].len[nb]);
                if (nb) bw.put((uint32_t)(diff < 0 ? diff - 1 : diff), nb);
                // AC run lengths
                int run = 0;
                int last = 63;
                while (last > 0 && q[last] == 0) last--;
                for (int i = 1; i <= last; i++) {
                    if (q[i] == 0) { run++; continue; }
                    while (run > 15) { bw.put(ac[t].code[0xf0], ac[t].len[0xf0]); run -= 16; }
                    nb = bitLength(q[i]);
                    const int sym = (run << 4) | nb;
                    bw.put(ac[t].code[sym], ac[t].len[sym]);
                    bw.put((uint32_t)(q[i] < 0 ? q[i] - 1 : q[i]), nb);
                    run = 0;
                }
                if (last < 63) bw.put(ac[t].code[0x00], ac[t].len[0x00]);   // EOB
            }
        }
    bw.flush();
    out.push_back(0xff); out.push_back(0xd9);
    std::ofstream f(path, std::ios::binary);
    if (!f) return false;
    f.write(reinterpret_cast<const char *>(out.data()), (std::streamsize)out.size());
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
