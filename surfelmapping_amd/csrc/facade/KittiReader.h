// KittiReader.h -- drop-in for the reference's dataset reader (gui/DatasetReader.h:16-75,
// gui/KittiReader.{h,cpp}) without OpenCV: reads <dir>/image_2/%06d.png (RGB), <dir>/PSMNet/%06d.png (u16 depth in
// millimetres), <dir>/semantics/%06d.png (u8 train ids), calibration.txt ("fx fy cx cy" / "width height"), pose.txt
// (3x4 row-major camera->world per line, right-multiplied by the -0.06 m x-offset T20: gui/KittiReader.cpp:296-302)
// and times.txt.  PNG decoding: 8/16-bit grey, 8-bit RGB/RGBA, non-interlaced, all five filter types, zlib inflate
// (link with -lz).  Only subLevel == 0 is supported (the reference's sub-sampling branch is unused by build_map.cpp:279
// and writes the sub-sampled semantics into the depth buffer, gui/KittiReader.cpp:206).
#pragma once
#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "sm_compat.h"

namespace sm_png {

struct Image { int w = 0, h = 0, channels = 0, depth = 0; std::vector<uint8_t> data; /* rows of w*channels*(depth/8), big-endian for 16 bit */ };

inline uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

inline bool read(const std::string &path, Image &img)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    std::vector<uint8_t> b((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (b.size() < 8 || std::memcmp(b.data(), sig, 8) != 0) return false;
    size_t pos = 8;
    std::vector<uint8_t> idat;
    int ctype = -1, interlace = 0;
    while (pos + 12 <= b.size()) {
        const uint32_t n = be32(&b[pos]);
        if (pos + 12 + n > b.size()) return false;
        const char *typ = (const char *)&b[pos + 4];
        const uint8_t *body = &b[pos + 8];
        if (!std::memcmp(typ, "IHDR", 4) && n >= 13) {
            img.w = (int)be32(body); img.h = (int)be32(body + 4); img.depth = body[8]; ctype = body[9]; interlace = body[12];
        } else if (!std::memcmp(typ, "IDAT", 4)) {
            idat.insert(idat.end(), body, body + n);
        } else if (!std::memcmp(typ, "IEND", 4)) {
            break;
        }
        pos += 12 + n;
    }
    if (img.w <= 0 || img.h <= 0 || interlace != 0) return false;
    img.channels = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 6 ? 4 : ctype == 4 ? 2 : 0;
    if (!img.channels || (img.depth != 8 && img.depth != 16)) return false;
    const size_t bpp = (size_t)img.channels * img.depth / 8, stride = (size_t)img.w * bpp;
    std::vector<uint8_t> raw((stride + 1) * img.h);
    uLongf outlen = (uLongf)raw.size();
    if (uncompress(raw.data(), &outlen, idat.data(), (uLong)idat.size()) != Z_OK || outlen != raw.size()) return false;
    img.data.assign(stride * img.h, 0);
    for (int y = 0; y < img.h; ++y) {
        const uint8_t ft = raw[(stride + 1) * y];
        const uint8_t *in = &raw[(stride + 1) * y + 1];
        uint8_t *out = &img.data[stride * y];
        const uint8_t *up = y ? &img.data[stride * (y - 1)] : nullptr;
        for (size_t i = 0; i < stride; ++i) {
            const int a = i >= bpp ? out[i - bpp] : 0, bb = up ? up[i] : 0, c = (up && i >= bpp) ? up[i - bpp] : 0;
            int pred = 0;
            switch (ft) {
                case 0: pred = 0; break;
                case 1: pred = a; break;
                case 2: pred = bb; break;
                case 3: pred = (a + bb) >> 1; break;
                case 4: { const int p = a + bb - c, pa = std::abs(p - a), pb = std::abs(p - bb), pc = std::abs(p - c);
                          pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? bb : c); break; }
                default: return false;
            }
            out[i] = (uint8_t)(in[i] + pred);
        }
    }
    return true;
}

}  // namespace sm_png

class KittiReader {
public:
    // gui/KittiReader.cpp:7-46
    KittiReader(std::string datasetDir, bool estimateDepth, bool useSemantic, int subLevel, bool groundTruth)
        : depth(nullptr), rgb(nullptr), semantic(nullptr), currentFrameId(-1), time(0.0), savedFrameId(-1),
          datasetDir_(std::move(datasetDir)), estimate_depth(estimateDepth), use_semantic(useSemantic), ok_(true)
    {
        if (subLevel != 0) { std::printf("KittiReader: subLevel != 0 is not supported\n"); ok_ = false; }
        std::ifstream timesIn(datasetDir_ + "/times.txt");
        double t;
        while (timesIn >> t) times.push_back(t);
        depthDir = datasetDir_ + "/PSMNet";
        rgbDir = datasetDir_ + "/image_2";
        semanticDir = datasetDir_ + "/semantics";
        ok_ = loadCalibration() && ok_;
        if (groundTruth) ok_ = loadGroundTruth() && ok_;
    }
    bool good() const { return ok_; }

    bool getNext() { ++currentFrameId; return load(); }                       // gui/KittiReader.cpp:56-75
    bool getLast() { --currentFrameId; return load(); }                       // :77-96
    void saveState() { savedFrameId = currentFrameId; }
    void resumeState() { currentFrameId = savedFrameId; }
    void setState(int frameId) { currentFrameId = frameId; }                  // next getNext() yields frameId + 1 (build_map.cpp:292)

    int W() const { return width_; }
    int H() const { return height_; }
    int numPixels() const { return width_ * height_; }
    float fx() const { return fx_; }
    float fy() const { return fy_; }
    float cx() const { return cx_; }
    float cy() const { return cy_; }
    const std::vector<Eigen::Matrix4f> *getGroundTruth() const { return &groundTruth_; }
    size_t numFrames() const { return times.size(); }

    unsigned short *depth;
    unsigned char *rgb;
    unsigned char *semantic;
    int currentFrameId;
    double time;
    int savedFrameId;
    Eigen::Matrix4f gtPose;

    // gui/KittiReader.cpp:239-280
    bool loadCalibration()
    {
        std::ifstream file(datasetDir_ + "/calibration.txt");
        std::string line;
        if (!file || !std::getline(file, line)) return false;
        if (std::sscanf(line.c_str(), "%f %f %f %f", &fx_, &fy_, &cx_, &cy_) != 4) return false;
        if (!std::getline(file, line)) return false;
        return std::sscanf(line.c_str(), "%d %d", &width_, &height_) == 2;
    }

    // gui/KittiReader.cpp:282-321: 3x4 row-major per line, times T20 (x offset -0.06 m)
    bool loadGroundTruth()
    {
        std::ifstream file(datasetDir_ + "/pose.txt");
        std::string line;
        groundTruth_.clear();
        while (std::getline(file, line)) {
            std::stringstream ss(line);
            float r[12];
            int n = 0;
            while (n < 12 && (ss >> r[n])) ++n;
            if (n != 12) continue;
            Eigen::Matrix4f g = Eigen::Matrix4f::Identity();
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 4; ++j) g(i, j) = r[4 * i + j];
            Eigen::Matrix4f out = g;                  // g * T20 with T20 = I + (-0.06) e_x e_w^T: only column 3 changes
            for (int i = 0; i < 3; ++i) out(i, 3) = g(i, 0) * -0.06f + g(i, 3);
            groundTruth_.push_back(out);
        }
        return groundTruth_.size() == times.size();
    }

private:
    bool load()
    {
        if (currentFrameId < 0 || (size_t)currentFrameId >= times.size()) return false;
        time = times[currentFrameId];
        if (!groundTruth_.empty()) gtPose = groundTruth_[currentFrameId];
        char name[32];
        std::snprintf(name, sizeof name, "/%06d.png", currentFrameId);
        const size_t P = (size_t)width_ * height_;
        sm_png::Image im;
        if (!sm_png::read(rgbDir + name, im) || im.w != width_ || im.h != height_ || im.depth != 8) {
            std::printf("CANNOT read RGB image from %s%s", rgbDir.c_str(), name);
            return false;
        }
        rgbBuf_.resize(P * 3);
        for (size_t p = 0; p < P; ++p)
            for (int c = 0; c < 3; ++c) rgbBuf_[p * 3 + c] = im.channels >= 3 ? im.data[p * im.channels + c] : im.data[p * im.channels];
        rgb = rgbBuf_.data();                      // R first, as after the reference's BGR->RGB swap (gui/KittiReader.cpp:131-134)
        if (!estimate_depth) {
            if (!sm_png::read(depthDir + name, im) || im.w != width_ || im.h != height_ || im.channels != 1) {
                std::printf("CANNOT read depth image from %s%s", depthDir.c_str(), name);
                return false;
            }
            depthBuf_.resize(P);
            for (size_t p = 0; p < P; ++p)
                depthBuf_[p] = im.depth == 16 ? (unsigned short)((im.data[p * 2] << 8) | im.data[p * 2 + 1]) : im.data[p];
            depth = depthBuf_.data();
        }
        if (use_semantic) {
            if (!sm_png::read(semanticDir + name, im) || im.w != width_ || im.h != height_ || im.channels != 1 || im.depth != 8) {
                std::printf("CANNOT read semantic image from %s%s", semanticDir.c_str(), name);
                return false;
            }
            semBuf_.assign(im.data.begin(), im.data.end());
            semantic = semBuf_.data();
        }
        return true;
    }

    const std::string datasetDir_;
    std::string depthDir, rgbDir, semanticDir;
    int width_ = 0, height_ = 0;
    float fx_ = 0, fy_ = 0, cx_ = 0, cy_ = 0;
    bool estimate_depth, use_semantic, ok_;
    std::vector<double> times;
    std::vector<Eigen::Matrix4f> groundTruth_;
    std::vector<unsigned char> rgbBuf_, semBuf_;
    std::vector<unsigned short> depthBuf_;
};
