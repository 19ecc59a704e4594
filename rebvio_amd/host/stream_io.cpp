// Stream sources / sinks (include/rebvio/io/stream_io.hpp). PNG decoding = chunk walk + zlib inflate + the five PNG
// scanline filters; nothing else of libpng is needed for camera frames.
#include "rebvio/io/stream_io.hpp"

#include <zlib.h>

#include <algorithm>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>

namespace rebvio {
namespace io {

namespace {
[[noreturn]] void bad(const std::string& what) { throw std::runtime_error(what); }

std::vector<unsigned char> slurp(const std::string& path) {
  std::ifstream f(path, std::ios::binary | std::ios::ate);
  if (!f) bad("cannot open " + path);
  const std::streamsize n = f.tellg();
  f.seekg(0);
  std::vector<unsigned char> b((size_t)n);
  if (n > 0 && !f.read(reinterpret_cast<char*>(b.data()), n)) bad("cannot read " + path);
  return b;
}
inline uint32_t be32(const unsigned char* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
inline int paeth(int a, int b, int c) {
  const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
  return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}
std::string trim(const std::string& s) {
  size_t a = 0, b = s.size();
  while (a < b && (s[a] == ' ' || s[a] == '\t' || s[a] == '\r' || s[a] == '\n')) ++a;
  while (b > a && (s[b - 1] == ' ' || s[b - 1] == '\t' || s[b - 1] == '\r' || s[b - 1] == '\n')) --b;
  return s.substr(a, b - a);
}
std::vector<std::string> csv_fields(const std::string& line) {
  std::vector<std::string> out;
  std::stringstream ss(line);
  std::string f;
  while (std::getline(ss, f, ',')) out.push_back(trim(f));
  return out;
}
}  // namespace

cv::Mat readPngGray(const std::string& path) {
  const std::vector<unsigned char> file = slurp(path);
  static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  if (file.size() < 8 + 25 || std::memcmp(file.data(), sig, 8) != 0) bad(path + ": not a PNG file");
  uint32_t W = 0, H = 0;
  int depth = 0, ctype = 0, interlace = 0;
  std::vector<unsigned char> idat;
  size_t pos = 8;
  bool end = false;
  while (!end && pos + 12 <= file.size()) {
    const uint32_t len = be32(&file[pos]);
    const char* type = reinterpret_cast<const char*>(&file[pos + 4]);
    if (pos + 12 + (size_t)len > file.size()) bad(path + ": truncated chunk");
    const unsigned char* data = &file[pos + 8];
    if (!std::memcmp(type, "IHDR", 4)) {
      if (len < 13) bad(path + ": bad IHDR");
      W = be32(data);
      H = be32(data + 4);
      depth = data[8];
      ctype = data[9];
      interlace = data[12];
    } else if (!std::memcmp(type, "IDAT", 4)) {
      idat.insert(idat.end(), data, data + len);
    } else if (!std::memcmp(type, "IEND", 4)) {
      end = true;
    }
    pos += 12 + (size_t)len;
  }
  if (W == 0 || H == 0 || W > 16384 || H > 16384) bad(path + ": bad size");
  if (interlace != 0) bad(path + ": interlaced PNG not supported");
  int channels;
  switch (ctype) {
    case 0: channels = 1; break;
    case 2: channels = 3; break;
    case 4: channels = 2; break;
    case 6: channels = 4; break;
    default: bad(path + ": palette PNG not supported");
  }
  if (!(depth == 8 || (depth == 16 && ctype == 0))) bad(path + ": unsupported bit depth");
  const size_t bpp = (size_t)channels * depth / 8, stride = (size_t)W * bpp;
  std::vector<unsigned char> raw((stride + 1) * H);
  uLongf out_len = (uLongf)raw.size();
  if (uncompress(raw.data(), &out_len, idat.data(), (uLong)idat.size()) != Z_OK || out_len != raw.size()) bad(path + ": inflate failed");
  std::vector<unsigned char> prev(stride, 0), cur(stride);
  cv::Mat img((int)H, (int)W, CV_8UC1);
  for (uint32_t y = 0; y < H; ++y) {
    const unsigned char ft = raw[(stride + 1) * y];
    const unsigned char* in = &raw[(stride + 1) * y + 1];
    for (size_t i = 0; i < stride; ++i) {
      const int a = i >= bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= bpp ? prev[i - bpp] : 0;
      int v = in[i];
      switch (ft) {
        case 0: break;
        case 1: v += a; break;
        case 2: v += b; break;
        case 3: v += (a + b) >> 1; break;
        case 4: v += paeth(a, b, c); break;
        default: bad(path + ": bad filter type");
      }
      cur[i] = (unsigned char)v;
    }
    unsigned char* o = img.ptr<unsigned char>((int)y);
    for (uint32_t x = 0; x < W; ++x) {
      const unsigned char* px = &cur[(size_t)x * bpp];
      if (ctype == 0 || ctype == 4) {
        o[x] = px[0];  // 16-bit grey: most significant byte
      } else {          // RGB -> luma with the fixed-point weights of cv::cvtColor(RGB2GRAY): (R*4899 + G*9617 + B*1868 + 8192) >> 14
        o[x] = (unsigned char)((px[0] * 4899 + px[1] * 9617 + px[2] * 1868 + 8192) >> 14);
      }
    }
    prev.swap(cur);
  }
  return img;
}

EurocReader::EurocReader(const std::string& mav0, const std::string& cam, const std::string& imu) {
  {
    const std::string dir = mav0 + "/" + cam;
    std::ifstream f(dir + "/data.csv");
    if (!f) bad("cannot open " + dir + "/data.csv");
    std::string line;
    while (std::getline(f, line)) {
      line = trim(line);
      if (line.empty() || line[0] == '#') continue;
      const auto fs = csv_fields(line);
      if (fs.size() < 2) continue;
      const uint64_t ns = std::stoull(fs[0]);
      frames_.push_back(FrameRef{ns / 1000ull, dir + "/data/" + fs[1], 0});
    }
    std::stable_sort(frames_.begin(), frames_.end(), [](const FrameRef& a, const FrameRef& b) { return a.ts_us < b.ts_us; });
  }
  {
    std::ifstream f(mav0 + "/" + imu + "/data.csv");
    if (f) {  // a camera-only dataset is allowed
      std::string line;
      while (std::getline(f, line)) {
        line = trim(line);
        if (line.empty() || line[0] == '#') continue;
        const auto fs = csv_fields(line);
        if (fs.size() < 7) continue;
        rebvio::types::Imu s;
        s.ts = std::stoull(fs[0]) / 1000ull;
        s.gyro = TooN::makeVector(std::stof(fs[1]), std::stof(fs[2]), std::stof(fs[3]));
        s.acc = TooN::makeVector(std::stof(fs[4]), std::stof(fs[5]), std::stof(fs[6]));
        imu_.push_back(s);
      }
      std::stable_sort(imu_.begin(), imu_.end(), [](const rebvio::types::Imu& a, const rebvio::types::Imu& b) { return a.ts < b.ts; });
    }
  }
}

RawReader::RawReader(const std::string& frames_file, int rows, int cols, uint64_t first_ts_us, uint64_t frame_dt_us,
                     const std::string& imu_file)
    : path_(frames_file), rows_(rows), cols_(cols), n_(0), t0_(first_ts_us), dt_(frame_dt_us) {
  std::ifstream f(frames_file, std::ios::binary | std::ios::ate);
  if (!f) bad("cannot open " + frames_file);
  n_ = (size_t)f.tellg() / ((size_t)rows * cols);
  if (!imu_file.empty()) {
    const std::vector<unsigned char> b = slurp(imu_file);
    for (size_t o = 0; o + 32 <= b.size(); o += 32) {
      int64_t ts;
      float g[3], a[3];
      std::memcpy(&ts, &b[o], 8);
      std::memcpy(g, &b[o + 8], 12);
      std::memcpy(a, &b[o + 20], 12);
      imu_.push_back(rebvio::types::Imu{(uint64_t)ts, TooN::makeVector(g[0], g[1], g[2]), TooN::makeVector(a[0], a[1], a[2])});
    }
  }
}

cv::Mat RawReader::frame(size_t i) {
  if (i >= n_) bad("RawReader: frame index out of range");
  cv::Mat m(rows_, cols_, CV_8UC1);
  std::ifstream f(path_, std::ios::binary);
  f.seekg((std::streamoff)(i * (size_t)rows_ * cols_));
  if (!f.read(reinterpret_cast<char*>(m.data), (std::streamsize)((size_t)rows_ * cols_))) bad("RawReader: short read");
  return m;
}

OdometryWriter::OdometryWriter(const std::string& path) : f_(std::fopen(path.c_str(), "w")) {
  if (!f_) bad("cannot create " + path);
}
OdometryWriter::~OdometryWriter() {
  if (f_) std::fclose(f_);
}
std::string OdometryWriter::format(const rebvio::types::Odometry& o) {
  char buf[256];
  std::snprintf(buf, sizeof(buf), "%llu %.6f %.6f %.6f %.6f %.6f %.6f", (unsigned long long)o.ts_us, (double)o.orientation[0],
                (double)o.orientation[1], (double)o.orientation[2], (double)o.position[0], (double)o.position[1], (double)o.position[2]);
  return buf;
}
void OdometryWriter::write(const rebvio::types::Odometry& o) {
  std::fprintf(f_, "%s\n", format(o).c_str());
  std::fflush(f_);
}

size_t replay(StreamSource& src, const std::function<void(rebvio::types::Image&&)>& image_cb,
              const std::function<void(rebvio::types::Imu&&)>& imu_cb, size_t first, size_t count) {
  const auto& imu = src.imu();
  size_t k = 0, delivered = 0;
  const size_t last = std::min(src.numFrames(), count == (size_t)-1 ? src.numFrames() : first + count);
  // samples older than the first played frame belong to nobody
  if (first < src.numFrames() && first > 0)
    while (k < imu.size() && imu[k].ts <= src.frameTs(first - 1)) ++k;
  for (size_t i = first; i < last; ++i) {
    const uint64_t ts = src.frameTs(i);
    for (; k < imu.size() && imu[k].ts <= ts; ++k)
      if (imu_cb) imu_cb(rebvio::types::Imu(imu[k]));
    image_cb(rebvio::types::Image{ts, src.frame(i)});
    ++delivered;
  }
  return delivered;
}

}  // namespace io
}  // namespace rebvio
