// lib_python.cpp -- drop-in replacement of the reference's pybind11 module `lib_python`
// (reference lib/PythonBindings.cpp:170-555) for the optimizer path.
//
// Host mirror of the reference's data model, as thin structs without OpenCV / Eigen / boost / glog:
//   DepthVideo / DepthStream / DepthFrame          reference lib/DepthVideo.{h,cpp}, lib/DepthStream.{h,cpp}
//   Extrinsics / Intrinsics / Quaternionf          reference lib/DepthPhoto.{h,cpp}:20-222
//   XformDescriptor / Xform / DepthXform / SpatialXform   reference lib/DepthMapTransform.{h,cpp}
//   FrameRange                                      reference lib/FrameRange.{h,cpp}
//   FlowConstraintsParams / FlowConstraintsCollection      reference lib/FlowConstraints.{h,cpp} (container, cache file, flags)
//   DepthVideoImporter::importVideo                 reference lib/Importer.cpp:25-38,197-238
//   DepthVideoPoseOptimizer::Params                 reference lib/PoseOptimizer.h:54-108
//   DepthVideoProcessor                             reference lib/Processor.{h,cpp} (the ops of this path)
// normalizeDepth / optimizePoses run on the MI355X through the C ABI of libcvd_hip.so (include/cvd_hip.h).
// The same Python names, argument meaning and error behaviour (std::runtime_error -> RuntimeError) as the
// reference, so that the reference's pose_optimization.py / params.py / loaders/video_dataset.py import it
// unchanged.  Out-of-scope pieces (COLMAP import, tracks, filters) raise; constraint sampling from the flow images runs on the device.
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <set>
#include <sstream>
#include <stdexcept>
#include <string>
#include <sys/stat.h>
#include <vector>

#include "../../include/cvd_hip.h"
#include "cvd_device.h"  // host+device gather helpers (one implementation of the spline taps)

namespace py = pybind11;

namespace cvdhost {

static bool fileExists(const std::string& p) {
  struct stat st;
  return ::stat(p.c_str(), &st) == 0 && S_ISREG(st.st_mode);
}
static bool dirExists(const std::string& p) {
  struct stat st;
  return ::stat(p.c_str(), &st) == 0 && S_ISDIR(st.st_mode);
}
static bool g_logStdout = false;
static bool g_denseHandOver = true;  // setDenseHandOver(): matchSeparation = 0 collections may go to the solver as images
static void logInfo(const std::string& s) {
  if (g_logStdout) { std::fputs(s.c_str(), stdout); std::fputc('\n', stdout); }
}
template <typename T>
static void wr(std::ostream& os, const T& v) { os.write(reinterpret_cast<const char*>(&v), sizeof(T)); }
template <typename T>
static T rd(std::istream& is) { T v{}; is.read(reinterpret_cast<char*>(&v), sizeof(T)); return v; }
static void wrstr(std::ostream& os, const std::string& s) {  // core/FileIo.cpp:175-181
  wr<size_t>(os, s.size());
  if (!s.empty()) os.write(s.data(), s.size());
}
static std::string fmtInt6(int v) { char b[32]; std::snprintf(b, sizeof(b), "%06d", v); return b; }

// ---- enums (same enumerator names / values as the reference) ----------------------------------------
enum class ValueXformType { None, Scale, ScaleShift };
enum class XformType { Depth, Spatial };
enum class DepthXformType { None, Identity, Global, Grid };
enum class SpatialXformType { None, Identity, VerticalLinear, CornersBilinear, BilinearGrid, BicubicGrid };
enum class StaticLossType { Euclidean, ReproDisparity, ReproDepthRatio, ReproLogDepth };
enum class SmoothLossType { EuclideanLaplacian, ReproDisparityLaplacian, ReproDepthRatioConsistency, ReproLogDepthConsistency };
enum class IntrinsicsOptimization { Fixed, Shared, PerFrame };

// ---- FrameRange (reference lib/FrameRange.cpp) --------------------------------------------------------
struct FrameRange {
  std::set<int> frames;
  void fromString(const std::string& str) {
    frames.clear();
    std::stringstream ss(str);
    std::string piece;
    while (std::getline(ss, piece, ',')) {
      if (piece.empty()) continue;
      const size_t dash = piece.find('-', 1);
      int start, end;
      if (dash == std::string::npos) {
        start = end = std::stoi(piece);
      } else {
        if (piece.find('-', dash + 1) != std::string::npos) throw std::runtime_error("Malformed range piece.");
        start = std::stoi(piece.substr(0, dash));
        end = std::stoi(piece.substr(dash + 1));
      }
      for (int f = start; f <= end; ++f) frames.insert(f);
    }
  }
  std::string toString() const {
    if (frames.empty()) return "";
    std::string res;
    auto it = frames.begin();
    int start = *it, last = start;
    auto add = [&]() {
      if (!res.empty()) res += ",";
      res += (last == start) ? std::to_string(start) : std::to_string(start) + "-" + std::to_string(last);
    };
    for (++it; it != frames.end(); ++it) {
      if (*it - last > 1) { add(); start = *it; }
      last = *it;
    }
    add();
    return res;
  }
  void resolve(int numFrames, bool clip = false) {
    if (clip) {
      std::set<int> c;
      for (int f : frames) if (f >= 0 && f < numFrames) c.insert(f);
      frames = c;
    }
    if (frames.empty()) for (int f = 0; f < numFrames; ++f) frames.insert(f);
    if (firstFrame() < 0 || lastFrame() >= numFrames)
      throw std::runtime_error("Frame range contains out-of-range frame indices.");
  }
  bool isEmpty() const { return frames.empty(); }
  void checkEmpty() const {
    if (frames.empty()) throw std::runtime_error("Frame set is empty. Forgot to call resolve()?");
  }
  int firstFrame() const { checkEmpty(); return *frames.begin(); }
  int lastFrame() const { checkEmpty(); return *frames.rbegin(); }
  int count() const { checkEmpty(); return static_cast<int>(frames.size()); }
  bool isConsecutive() const { checkEmpty(); return (lastFrame() - firstFrame() + 1) == static_cast<int>(frames.size()); }
  bool inRange(int f) const { checkEmpty(); return frames.count(f) != 0; }
};

// ---- Extrinsics / Intrinsics (reference lib/DepthPhoto.cpp:39-157) -------------------------------------
struct Quaternionf {
  float x_ = 0.f, y_ = 0.f, z_ = 0.f, w_ = 1.f;
  std::array<float, 3> rotate(const std::array<float, 3>& v) const {  // Eigen: v + w*uv + qv x uv, uv = 2 qv x v
    const float uv[3] = {2.f * (y_ * v[2] - z_ * v[1]), 2.f * (z_ * v[0] - x_ * v[2]), 2.f * (x_ * v[1] - y_ * v[0])};
    return {v[0] + w_ * uv[0] + (y_ * uv[2] - z_ * uv[1]), v[1] + w_ * uv[1] + (z_ * uv[0] - x_ * uv[2]),
            v[2] + w_ * uv[2] + (x_ * uv[1] - y_ * uv[0])};
  }
};
struct Extrinsics {
  std::array<float, 3> position{0.f, 0.f, 0.f};
  Quaternionf orientation;
  std::array<float, 3> left() const { return orientation.rotate({-1.f, 0.f, 0.f}); }
  std::array<float, 3> right() const { return orientation.rotate({1.f, 0.f, 0.f}); }
  std::array<float, 3> down() const { return orientation.rotate({0.f, -1.f, 0.f}); }
  std::array<float, 3> up() const { return orientation.rotate({0.f, 1.f, 0.f}); }
  std::array<float, 3> forward() const { return orientation.rotate({0.f, 0.f, -1.f}); }
  std::array<float, 3> backward() const { return orientation.rotate({0.f, 0.f, 1.f}); }
};
struct Intrinsics {
  static constexpr float kDefaultHFov = 0.508015513f, kDefaultVFov = 0.666488587f;
  int projection = 0;
  float vFov = 0.f, hFov = 0.f, centerLat = 0.f, centerLon = 0.f;
  void resolveMissingFov(float aspect) {
    bool vSet = vFov > 0, hSet = hFov > 0;
    if (vSet && hSet) return;
    if (aspect == 0) throw std::runtime_error("Aspect ratio must be non-zero.");
    const float defAspect = tanf(kDefaultHFov / 2.f) / tanf(kDefaultVFov / 2.f);
    if (!vSet && !hSet) {
      if (aspect > defAspect) { vFov = kDefaultVFov; vSet = true; } else { hFov = kDefaultHFov; hSet = true; }
    }
    if (vSet) hFov = std::atan(std::tan(vFov / 2.0f) * aspect) * 2.0f;
    else if (hSet) vFov = std::atan(std::tan(hFov / 2.0f) / aspect) * 2.0f;
  }
};

// ---- XformDescriptor / Xform (reference lib/DepthMapTransform.cpp) ---------------------------------------
static const char* kValueStr[] = {"None", "Scale", "ScaleShift"};
static const char* kDepthStr[] = {"None", "Identity", "Global", "Grid"};
static const char* kSpatialStr[] = {"None", "Identity", "VerticalLinear", "CornersBilinear", "BilinearGrid", "BicubicGrid"};
template <typename E, size_t N>
static E parseEnum(const std::string& s, const char* (&names)[N]) {
  for (size_t i = 0; i < N; ++i) if (s == names[i]) return static_cast<E>(i);
  throw std::runtime_error("Invalid enum value '" + s + "'.");
}
static std::string trim(std::string s) {
  const char* ws = " \t\r\n";
  s.erase(0, s.find_first_not_of(ws));
  s.erase(s.find_last_not_of(ws) + 1);
  return s;
}

struct XformDescriptor {
  XformType type = XformType::Depth;
  DepthXformType depthType = DepthXformType::Identity;
  SpatialXformType spatialType = SpatialXformType::None;
  ValueXformType valueXform = ValueXformType::None;
  bool cubicInterpolation = false;  // not bound by the reference; exposed here as an extension
  std::array<int, 3> gridSize{0, 0, 0};
  std::array<double, 2> depthMinMax{0.0, 0.0};

  void reset(XformType t = XformType::Depth) {  // :106-114
    *this = XformDescriptor();
    if (t == XformType::Spatial) {
      type = XformType::Spatial;
      depthType = DepthXformType::None;
      spatialType = SpatialXformType::Identity;
    }
  }
  std::string str() const {  // :116-165
    char buf[256];
    if (type == XformType::Depth) {
      std::string res = std::string(kDepthStr[static_cast<int>(depthType)]) + "(";
      const char* v = kValueStr[static_cast<int>(valueXform)];
      switch (depthType) {
        case DepthXformType::Identity: break;
        case DepthXformType::Global: res += v; break;
        case DepthXformType::Grid:
          if (gridSize[2] > 1)
            std::snprintf(buf, sizeof(buf), "%s, %s, %d, %d, %d, %f, %f", v, cubicInterpolation ? "Cubic" : "Linear",
                          gridSize[0], gridSize[1], gridSize[2], depthMinMax[0], depthMinMax[1]);
          else
            std::snprintf(buf, sizeof(buf), "%s, %s, %d, %d, %d", v, cubicInterpolation ? "Cubic" : "Linear",
                          gridSize[0], gridSize[1], gridSize[2]);
          res += buf;
          break;
        default: throw std::runtime_error("Invalid depth transform type.");
      }
      return res + ")";
    }
    std::string res = kSpatialStr[static_cast<int>(spatialType)];
    if (spatialType == SpatialXformType::BilinearGrid || spatialType == SpatialXformType::BicubicGrid) {
      std::snprintf(buf, sizeof(buf), "(%d, %d)", gridSize[0], gridSize[1]);
      res += buf;
    }
    return res;
  }
  void parse(const std::string& s) {  // :167-265
    depthType = DepthXformType::None;
    spatialType = SpatialXformType::None;
    const size_t pos = s.find('(');
    const std::string typeStr = s.substr(0, pos);
    std::vector<std::string> args;
    auto getArgs = [&]() {
      if (pos == std::string::npos || s.back() != ')') throw std::runtime_error("Malformed descriptor string.");
      std::stringstream ss(s.substr(pos + 1, s.size() - 1 - (pos + 1)));
      std::string a;
      while (std::getline(ss, a, ',')) args.push_back(trim(a));
    };
    if (type == XformType::Depth) {
      getArgs();
      if (typeStr == "BicubicGrid" || typeStr == "BilinearGrid") {
        args = {args.at(0), typeStr == "BicubicGrid" ? "Cubic" : "Linear", args.at(1), args.at(2), "1"};
        depthType = DepthXformType::Grid;
      } else {
        depthType = parseEnum<DepthXformType>(typeStr, kDepthStr);
      }
      switch (depthType) {
        case DepthXformType::Identity:
          if (!args.empty()) throw std::runtime_error("Incorrect number of parameters.");
          break;
        case DepthXformType::Global:
          if (args.size() != 1) throw std::runtime_error("Incorrect number of parameters.");
          valueXform = parseEnum<ValueXformType>(args[0], kValueStr);
          break;
        case DepthXformType::Grid:
          if (args.size() < 5) throw std::runtime_error("Incorrect number of parameters.");
          valueXform = parseEnum<ValueXformType>(args[0], kValueStr);
          if (args[1] == "Cubic") cubicInterpolation = true;
          else if (args[1] == "Linear") cubicInterpolation = false;
          else throw std::runtime_error("Invalid interpolation mode.");
          gridSize = {std::stoi(args[2]), std::stoi(args[3]), std::stoi(args[4])};
          if (gridSize[2] <= 1) {
            if (args.size() != 5) throw std::runtime_error("Incorrect number of parameters.");
          } else {
            if (args.size() != 7) throw std::runtime_error("Incorrect number of parameters.");
            depthMinMax = {std::stof(args[5]), std::stof(args[6])};
          }
          break;
        default: throw std::runtime_error("Invalid depth transform type.");
      }
    } else {
      spatialType = parseEnum<SpatialXformType>(typeStr, kSpatialStr);
      if (spatialType == SpatialXformType::BilinearGrid || spatialType == SpatialXformType::BicubicGrid) {
        getArgs();
        if (args.size() != 2) throw std::runtime_error("Incorrect number of parameters.");
        gridSize[0] = std::stoi(args[0]);
        gridSize[1] = std::stoi(args[1]);
      }
    }
  }
  void fwrite(std::ostream& os) const {  // :276-279
    wr<int32_t>(os, static_cast<int32_t>(type));
    wrstr(os, str());
  }
  bool operator==(const XformDescriptor& o) const {  // :313-323 (ignores cubic flag and depthMinMax)
    return type == o.type && depthType == o.depthType && spatialType == o.spatialType && valueXform == o.valueXform &&
           gridSize == o.gridSize;
  }
  cvd_xform_desc toC() const {
    cvd_xform_desc d{};
    d.type = static_cast<int>(type);
    d.depth_type = static_cast<int>(depthType);
    d.spatial_type = static_cast<int>(spatialType);
    d.value_xform = static_cast<int>(valueXform);
    d.cubic_interpolation = cubicInterpolation ? 1 : 0;
    for (int i = 0; i < 3; ++i) d.grid_size[i] = gridSize[i];
    d.depth_min_max[0] = depthMinMax[0];
    d.depth_min_max[1] = depthMinMax[1];
    return d;
  }
  static XformDescriptor fromC(const cvd_xform_desc& d) {
    XformDescriptor x;
    x.type = static_cast<XformType>(d.type);
    x.depthType = static_cast<DepthXformType>(d.depth_type);
    x.spatialType = static_cast<SpatialXformType>(d.spatial_type);
    x.valueXform = static_cast<ValueXformType>(d.value_xform);
    x.cubicInterpolation = d.cubic_interpolation != 0;
    x.gridSize = {d.grid_size[0], d.grid_size[1], d.grid_size[2]};
    x.depthMinMax = {d.depth_min_max[0], d.depth_min_max[1]};
    return x;
  }
};

static int valueNumParams(ValueXformType t) {
  if (t == ValueXformType::Scale) return 1;
  if (t == ValueXformType::ScaleShift) return 2;
  throw std::runtime_error("Invalid value transform.");
}

struct DepthFrame;

struct Xform {
  XformDescriptor desc_;
  std::vector<double> params_;
  virtual ~Xform() = default;
  const XformDescriptor& desc() const { return desc_; }
  const std::vector<double>& params() const { return params_; }
  int numParams() const { return static_cast<int>(params_.size()); }
  std::string str() const {
    std::string res = desc_.str() + " [";
    char b[64];
    for (size_t i = 0; i < params_.size(); ++i) {
      std::snprintf(b, sizeof(b), "%s%.2f", i ? ", " : "", params_[i]);
      res += b;
    }
    return res + "]";
  }
  void copyFrom(const Xform& o) {
    if (!(o.desc_ == desc_)) throw std::runtime_error("Can only copy parameters from same type of transform.");
    params_ = o.params_;
  }
  int blockSize() const {
    if (desc_.type == XformType::Depth) return desc_.depthType == DepthXformType::Identity ? 0 : valueNumParams(desc_.valueXform);
    return desc_.spatialType == SpatialXformType::Identity ? 0 : 2;
  }
  // taps of one sample (depth or spatial), using the shared host/device spline helpers
  // srcDepth: the sample's source depth, looked at by depth-wise grids only (gridSize.z > 1)
  int gather(float lx, float ly, int* idx, double* w, float srcDepth = 0.f) const {
    const int gx = desc_.gridSize[0], gy = desc_.gridSize[1];
    const double mx = std::nextafter(static_cast<double>(gx - 1), 0.0), my = std::nextafter(static_cast<double>(gy - 1), 0.0);
    if (desc_.type == XformType::Depth) {
      switch (desc_.depthType) {
        case DepthXformType::Identity: return 0;
        case DepthXformType::Global: idx[0] = 0; w[0] = 1.0; return 1;
        case DepthXformType::Grid:
          if (desc_.gridSize[2] > 1) {
            // GridDepthXform::linearGather with a depth-wise axis, reference lib/DepthMapTransform.cpp:771-851
            if (desc_.cubicInterpolation) throw std::runtime_error("Cubic interpolation of depth-wise grids is not defined.");
            const int gz = desc_.gridSize[2];
            const double dmin = 1.0 / desc_.depthMinMax[1], dmax = 1.0 / desc_.depthMinMax[0];
            const double interval = (dmax - dmin) / (gz - 1), mz = std::nextafter(static_cast<double>(gz - 1), 0.0);
            double sz = (1.0 / static_cast<double>(srcDepth) - dmin) / interval;
            sz = sz < 0.0 ? 0.0 : (sz > mz ? mz : sz);
            const int iz = static_cast<int>(sz);
            const double rz = sz - iz;
            if (gx > 1) {
              int i4[4];
              double w4[4];
              cvd::bilinearTaps(lx, ly, gx, gy, mx, my, i4, w4);
              for (int k = 0; k < 4; ++k) {
                idx[k] = i4[k] + iz * gx * gy;           w[k] = w4[k] * (1.0 - rz);
                idx[4 + k] = i4[k] + (iz + 1) * gx * gy; w[4 + k] = w4[k] * rz;
              }
              return 8;
            }
            idx[0] = iz; w[0] = 1.0 - rz;
            idx[1] = iz + 1; w[1] = rz;
            return 2;
          }
          if (desc_.cubicInterpolation) return cvd::bicubicTaps(lx, ly, gx, gy, mx, my, idx, w);
          cvd::bilinearTaps(lx, ly, gx, gy, mx, my, idx, w);
          return 4;
        default: throw std::runtime_error("Invalid depth transform type.");
      }
    }
    switch (desc_.spatialType) {
      case SpatialXformType::Identity: return 0;
      case SpatialXformType::VerticalLinear: {
        const double w0 = 0.5 + 0.5 * ly;
        idx[0] = 0; w[0] = w0; idx[1] = 1; w[1] = 1.0 - w0;
        return 2;
      }
      case SpatialXformType::CornersBilinear: {
        const double wx = 0.5 + 0.5 * lx, wy = 0.5 + 0.5 * ly;
        idx[0] = 0; w[0] = wx * wy; idx[1] = 1; w[1] = (1.0 - wx) * wy;
        idx[2] = 2; w[2] = wx * (1.0 - wy); idx[3] = 3; w[3] = (1.0 - wx) * (1.0 - wy);
        return 4;
      }
      case SpatialXformType::BilinearGrid: cvd::bilinearTaps(lx, ly, gx, gy, mx, my, idx, w); return 4;
      case SpatialXformType::BicubicGrid: return cvd::bicubicTaps(lx, ly, gx, gy, mx, my, idx, w);
      default: throw std::runtime_error("Invalid spatial transform type.");
    }
  }
};

static int xformNumParams(const XformDescriptor& d) {
  if (d.type == XformType::Depth) {
    switch (d.depthType) {
      case DepthXformType::Identity: return 0;
      case DepthXformType::Global: return valueNumParams(d.valueXform);
      case DepthXformType::Grid: {
        if (d.gridSize[0] > 1 || d.gridSize[1] > 1)
          if (d.gridSize[0] < 2 || d.gridSize[1] < 2)
            throw std::runtime_error("Spatial grid transforms must have at least two rows and columns, respectively.");
        const int n = valueNumParams(d.valueXform) * d.gridSize[0] * d.gridSize[1] * d.gridSize[2];
        if (n <= 1) throw std::runtime_error("Grid transform cannot have an empty grid.");
        return n;
      }
      default: throw std::runtime_error("Invalid depth transform type.");
    }
  }
  switch (d.spatialType) {
    case SpatialXformType::Identity: return 0;
    case SpatialXformType::VerticalLinear: return 4;
    case SpatialXformType::CornersBilinear: return 8;
    case SpatialXformType::BilinearGrid:
    case SpatialXformType::BicubicGrid:
      if (d.gridSize[1] < 2 || d.gridSize[0] < 2)
        throw std::logic_error("Need at least two rows and columns in depth transform grid.");
      return d.gridSize[0] * d.gridSize[1] * 2;
    default: throw std::runtime_error("Invalid spatial transform type.");
  }
}

struct DepthXform : Xform {
  py::array paramMap(const DepthFrame& df) const;  // reference :950-994 (grid only)
  std::vector<float> apply(const std::vector<float>& src, int w, int h) const {  // reference :394-415
    std::vector<float> dst(src.size());
    const float xs = 2.f / (w - 1.f), ys = 2.f / (h - 1.f);
    const int N = blockSize();
    int idx[16];
    double wt[16];
    for (int y = 0; y < h; ++y) {
      const float ly = 1.f - y * ys;
      for (int x = 0; x < w; ++x) {
        const float lx = -1.f + x * xs;
        const double d = src[static_cast<size_t>(y) * w + x];
        if (N == 0) { dst[static_cast<size_t>(y) * w + x] = static_cast<float>(d); continue; }
        const int n = gather(lx, ly, idx, wt, src[static_cast<size_t>(y) * w + x]);
        double D = 0.0;
        for (int k = 0; k < n; ++k)
          D += ((N == 2) ? (d * params_[idx[k] * 2] + params_[idx[k] * 2 + 1]) : d * params_[idx[k]]) * wt[k];
        dst[static_cast<size_t>(y) * w + x] = static_cast<float>(D);
      }
    }
    return dst;
  }
};

struct SpatialXform : Xform {
  py::array_t<float> warp(int h, int w) const {  // reference :428-449
    py::array_t<float> out({h, w, 2});
    auto a = out.mutable_unchecked<3>();
    const float xs = 2.f / (w - 1.f), ys = 2.f / (h - 1.f);
    int idx[16];
    double wt[16];
    for (int y = 0; y < h; ++y) {
      const float ly = 1.f - y * ys;
      for (int x = 0; x < w; ++x) {
        const float lx = -1.f + x * xs;
        const int n = gather(lx, ly, idx, wt);
        double wx = 0.0, wy = 0.0;
        for (int k = 0; k < n; ++k) { wx += params_[idx[k] * 2] * wt[k]; wy += params_[idx[k] * 2 + 1] * wt[k]; }
        a(y, x, 0) = static_cast<float>(wx);
        a(y, x, 1) = static_cast<float>(wy);
      }
    }
    return out;
  }
};

static std::unique_ptr<DepthXform> createDepthXform(const XformDescriptor& d) {
  if (d.type != XformType::Depth) throw std::runtime_error("Transform has the wrong type.");
  auto x = std::make_unique<DepthXform>();
  x->desc_ = d;
  x->params_.assign(xformNumParams(d), 1.0);
  return x;
}
static std::unique_ptr<SpatialXform> createSpatialXform(const XformDescriptor& d) {
  if (d.type != XformType::Spatial) throw std::runtime_error("Transform has the wrong type.");
  auto x = std::make_unique<SpatialXform>();
  x->desc_ = d;
  x->params_.assign(xformNumParams(d), 0.0);
  return x;
}
static void writeXform(std::ostream& os, const Xform& x) {  // :1501-1506
  x.desc_.fwrite(os);
  os.write(reinterpret_cast<const char*>(x.params_.data()), sizeof(double) * x.params_.size());
}

// ---- DepthVideo data model -------------------------------------------------------------------------------
struct DepthVideo;
struct DepthStream;

struct DepthFrame {
  DepthFrame() = default;
  DepthFrame(const DepthFrame&) = delete;
  DepthFrame& operator=(const DepthFrame&) = delete;
  DepthVideo* video = nullptr;
  DepthStream* stream = nullptr;
  int index = 0;
  Intrinsics intrinsics;
  Extrinsics extrinsics;
  bool enabled = true;
  std::unique_ptr<DepthXform> depthXform_;
  std::unique_ptr<SpatialXform> spatialXform_;
  std::vector<float> sourceDepth_;  // rows*cols, depth (not disparity); empty = not loaded
  bool triedLoad = false;

  DepthXform& depthXform() {
    if (!depthXform_) throw std::runtime_error("Depth transform not initialized.");
    return *depthXform_;
  }
  SpatialXform& spatialXform() {
    if (!spatialXform_) throw std::runtime_error("Spatial transform not initialized.");
    return *spatialXform_;
  }
  void resetDepthXform();
  void resetSpatialXform();
  const std::vector<float>* sourceDepth();  // lazy load of depth/frame_%06d.raw, disparity -> depth
  void setDepth(const py::array_t<float, py::array::c_style | py::array::forcecast>& d);
  void clearCache() { sourceDepth_.clear(); triedLoad = false; }
  void clearXformedCache() {}
  void clear() { clearCache(); intrinsics = Intrinsics(); extrinsics = Extrinsics(); }
  int width() const;
  int height() const;
};

struct DepthStream {
  DepthStream() = default;
  DepthStream(const DepthStream&) = delete;
  DepthStream& operator=(const DepthStream&) = delete;
  DepthVideo* video = nullptr;
  std::string name_, dir_, path_;
  XformDescriptor depthXformDesc_, spatialXformDesc_;
  int width_ = -1, height_ = -1;
  std::vector<std::unique_ptr<DepthFrame>> frames_;
  DepthFrame& frame(int i) { return *frames_.at(i); }
  void setDir(const std::string& dir);
  void initDimensions() {
    if (width_ >= 0) return;
    for (auto& f : frames_) if (f->sourceDepth()) return;
    width_ = height_ = 0;
  }
  int width() { if (width_ < 0) initDimensions(); return width_; }
  int height() { if (height_ < 0) initDimensions(); return height_; }
  void resetDepthXforms(const XformDescriptor& d) {
    depthXformDesc_ = d;
    for (auto& f : frames_) f->resetDepthXform();
  }
  void resetSpatialXforms(const XformDescriptor& d) {
    spatialXformDesc_ = d;
    for (auto& f : frames_) f->resetSpatialXform();
  }
  void clearCache() { for (auto& f : frames_) f->clearCache(); }
};

struct ColorStream {
  DepthVideo* video = nullptr;
  std::string name_, dir_, path_, extension_;
  int type_ = 0, width_ = -1, height_ = -1;
  void setDir(const std::string& dir);
};

struct DepthVideo {
  DepthVideo() = default;
  DepthVideo(const DepthVideo&) = delete;
  DepthVideo& operator=(const DepthVideo&) = delete;
  static constexpr uint32_t kFileFormatVersion = 13;
  std::string path_;
  std::vector<float> pts_;
  std::vector<std::unique_ptr<ColorStream>> colorStreams_;
  std::vector<std::unique_ptr<DepthStream>> depthStreams_;
  int width_ = 0, height_ = 0;
  float aspect_ = 1.f, invAspect_ = 1.f, duration_ = 0.f;

  void reset() {
    path_.clear(); pts_.clear(); colorStreams_.clear(); depthStreams_.clear();
    width_ = height_ = 0; aspect_ = invAspect_ = 1.f; duration_ = 0.f;
  }
  void init(const std::string& path, int w, int h, std::vector<float> pts) {  // reference lib/DepthVideo.cpp:103-119
    reset();
    path_ = path;
    pts_ = std::move(pts);
    width_ = w;
    height_ = h;
    aspect_ = w / float(h);
    invAspect_ = 1.f / aspect_;
    duration_ = pts_.size() > 1 ? pts_.back() * pts_.size() / float(pts_.size() - 1) : 0.f;
  }
  int numFrames() const { return static_cast<int>(pts_.size()); }
  int numColorStreams() const { return static_cast<int>(colorStreams_.size()); }
  int numDepthStreams() const { return static_cast<int>(depthStreams_.size()); }
  bool hasColorStream(const std::string& n) const {
    for (auto& c : colorStreams_) if (c->name_ == n) return true;
    return false;
  }
  int colorStreamIndex(const std::string& n) const {
    for (size_t i = 0; i < colorStreams_.size(); ++i) if (colorStreams_[i]->name_ == n) return static_cast<int>(i);
    throw std::runtime_error("Could not find named color stream.");
  }
  bool hasDepthStream(const std::string& n) const {
    for (auto& d : depthStreams_) if (d->name_ == n) return true;
    return false;
  }
  int depthStreamIndex(const std::string& n) const {
    for (size_t i = 0; i < depthStreams_.size(); ++i) if (depthStreams_[i]->name_ == n) return static_cast<int>(i);
    throw std::runtime_error("Could not find named depth stream.");
  }
  void createColorStream(const std::string& name, const std::string& dir, const std::string& ext, int type,
                         const std::pair<int, int>& size) {
    // OpenCV type codes: CV_8UC1 = 0, CV_8UC3 = 16, CV_32FC1 = 5, CV_32FC3 = 21 (reference :455-459)
    if (type != 0 && type != 16 && type != 5 && type != 21)
      throw std::runtime_error("Color streams only support 1 or 3 channels and byte or float depth.");
    colorStreams_.push_back(std::make_unique<ColorStream>());
    ColorStream& cs = *colorStreams_.back();
    cs.video = this;
    cs.name_ = name;
    cs.setDir(dir);
    cs.extension_ = ext;
    cs.type_ = type;
    cs.width_ = size.first;
    cs.height_ = size.second;
  }
  void createDepthStream(const std::string& name, const std::string& dir, const std::pair<int, int>& size) {
    depthStreams_.push_back(std::make_unique<DepthStream>());
    DepthStream& ds = *depthStreams_.back();
    ds.video = this;
    ds.name_ = name;
    ds.setDir(dir);
    ds.depthXformDesc_.reset();
    ds.spatialXformDesc_.reset(XformType::Spatial);
    ds.width_ = size.first;
    ds.height_ = size.second;
    for (int f = 0; f < numFrames(); ++f) {
      auto df = std::make_unique<DepthFrame>();
      df->video = this;
      df->stream = &ds;
      df->index = f;
      df->depthXform_ = createDepthXform(ds.depthXformDesc_);
      df->spatialXform_ = createSpatialXform(ds.spatialXformDesc_);
      df->intrinsics.resolveMissingFov(aspect_);
      ds.frames_.push_back(std::move(df));
    }
  }
  void clearDepthCaches() { for (auto& d : depthStreams_) d->clearCache(); }
  void printInfo() const {
    char b[512];
    logInfo("Path: " + path_);
    std::snprintf(b, sizeof(b), "Dimensions: %d x %d (%f aspect ratio)", width_, height_, aspect_);
    logInfo(b);
    std::snprintf(b, sizeof(b), "Frame count: %d (%.2fs duration)", numFrames(), duration_);
    logInfo(b);
    logInfo("Color streams: " + std::to_string(numColorStreams()));
    for (auto& c : colorStreams_) logInfo("  " + c->name_ + " (" + c->dir_ + ")");
    logInfo("Depth streams: " + std::to_string(numDepthStreams()));
    for (auto& d : depthStreams_) logInfo("  " + d->name_ + " (" + d->dir_ + "), " + d->depthXformDesc_.str() + ", " + d->spatialXformDesc_.str());
  }
  void save() const {  // video.dat, reference lib/DepthVideo.cpp:300-385 (format 13)
    std::ofstream os(path_ + "/video.dat", std::ios::binary);
    if (!os) throw std::runtime_error("Could not open 'video.dat' for writing.");
    wr<uint32_t>(os, 0xDEADBEEF);
    wr<uint32_t>(os, kFileFormatVersion);
    wr<uint32_t>(os, 3);  // DepthPhoto::kFileFormatVersion
    wr<int32_t>(os, numFrames());
    for (float p : pts_) wr<float>(os, p);
    wr<int32_t>(os, numColorStreams());
    for (auto& cs : colorStreams_) {
      wrstr(os, cs->name_); wrstr(os, cs->dir_); wrstr(os, cs->extension_);
      wr<int32_t>(os, cs->type_); wr<int32_t>(os, cs->width_); wr<int32_t>(os, cs->height_);
      wr<uint8_t>(os, 0);  // no GOP table
    }
    wr<int32_t>(os, numDepthStreams());
    for (auto& ds : depthStreams_) {
      wrstr(os, ds->name_); wrstr(os, ds->dir_);
      ds->depthXformDesc_.fwrite(os);
      ds->spatialXformDesc_.fwrite(os);
      wr<int32_t>(os, ds->width_); wr<int32_t>(os, ds->height_);
      wr<uint8_t>(os, 0);
      for (auto& f : ds->frames_) {
        wr<int32_t>(os, f->intrinsics.projection);
        wr<float>(os, f->intrinsics.vFov); wr<float>(os, f->intrinsics.hFov);
        wr<float>(os, f->intrinsics.centerLat); wr<float>(os, f->intrinsics.centerLon);
        for (float v : f->extrinsics.position) wr<float>(os, v);
        wr<float>(os, f->extrinsics.orientation.x_); wr<float>(os, f->extrinsics.orientation.y_);
        wr<float>(os, f->extrinsics.orientation.z_); wr<float>(os, f->extrinsics.orientation.w_);
        wr<uint8_t>(os, f->enabled ? 1 : 0);
        writeXform(os, *f->depthXform_);
        writeXform(os, *f->spatialXform_);
      }
    }
    wr<float>(os, duration_); wr<int32_t>(os, width_); wr<int32_t>(os, height_);
    wr<float>(os, aspect_); wr<float>(os, invAspect_);
    wr<uint32_t>(os, 0xDEADBEEF);
  }
};

void DepthStream::setDir(const std::string& dir) { dir_ = dir; path_ = video->path_ + "/" + dir_; }
void ColorStream::setDir(const std::string& dir) { dir_ = dir; path_ = video->path_ + "/" + dir_; }
void DepthFrame::resetDepthXform() { depthXform_ = createDepthXform(stream->depthXformDesc_); }
void DepthFrame::resetSpatialXform() { spatialXform_ = createSpatialXform(stream->spatialXformDesc_); }
int DepthFrame::width() const { return stream->width(); }
int DepthFrame::height() const { return stream->height(); }

static void checkDims(DepthStream& s, int w, int h) {  // reference lib/DepthStream.cpp:305-318
  if (s.width_ <= 0) { s.width_ = w; s.height_ = h; }
  else if (w != s.width_ || h != s.height_) throw std::runtime_error("Depth frame dimensions do not match stream dimensions.");
}

const std::vector<float>* DepthFrame::sourceDepth() {  // reference lib/DepthStream.cpp:176-216
  if (sourceDepth_.empty() && !triedLoad) {
    triedLoad = true;
    const std::string fn = stream->path_ + "/depth/frame_" + fmtInt6(index) + ".raw";
    std::ifstream is(fn, std::ios::binary);
    if (is) {
      // raw image header (reference lib/core/CvUtil.cpp:25-36): int rows, int cols, int cvType, size_t elemSize
      const int rows = rd<int32_t>(is), cols = rd<int32_t>(is), type = rd<int32_t>(is);
      const size_t elem = rd<size_t>(is);
      if (type != 5 || elem != 4 || rows <= 0 || cols <= 0) throw std::runtime_error("Unexpected depth image type in '" + fn + "'.");
      sourceDepth_.resize(static_cast<size_t>(rows) * cols);
      is.read(reinterpret_cast<char*>(sourceDepth_.data()), sourceDepth_.size() * 4);
      for (float& d : sourceDepth_) d = (std::isfinite(d) && d > 0.f) ? 1.f / d : 0.f;  // disparity -> depth
      checkDims(*stream, cols, rows);
    }
  }
  return sourceDepth_.empty() ? nullptr : &sourceDepth_;
}
void DepthFrame::setDepth(const py::array_t<float, py::array::c_style | py::array::forcecast>& d) {
  if (d.ndim() != 2) throw std::runtime_error("Depth must be a 2-D float array (H, W).");
  checkDims(*stream, static_cast<int>(d.shape(1)), static_cast<int>(d.shape(0)));
  sourceDepth_.assign(d.data(), d.data() + d.size());
  triedLoad = true;
}

py::array DepthXform::paramMap(const DepthFrame& dfc) const {  // reference lib/DepthMapTransform.cpp:950-994
  if (desc_.depthType != DepthXformType::Grid)
    throw std::runtime_error("Parameter map not implemented for this transform type.");
  DepthFrame& df = const_cast<DepthFrame&>(dfc);
  const int w = df.width(), h = df.height();
  const int N = blockSize();
  std::vector<ssize_t> shape = {h, w};
  if (N > 1) shape.push_back(N);
  py::array_t<double> out(shape);
  double* dst = out.mutable_data();
  const float xs = 2.f / (w - 1.f), ys = 2.f / (h - 1.f);
  int idx[16];
  double wt[16];
  const std::vector<float>* src = desc_.gridSize[2] > 1 ? df.sourceDepth() : nullptr;  // (depth-wise grids: reference :953,966)
  if (desc_.gridSize[2] > 1 && !src) throw std::runtime_error("Missing depth image.");
  for (int y = 0; y < h; ++y) {
    const float ly = 1.f - y * ys;
    for (int x = 0; x < w; ++x) {
      const float lx = -1.f + x * xs;
      const int n = gather(lx, ly, idx, wt, src ? (*src)[static_cast<size_t>(y) * w + x] : 0.f);
      double* o = dst + (static_cast<size_t>(y) * w + x) * N;
      for (int d = 0; d < N; ++d) o[d] = 0.0;
      for (int k = 0; k < n; ++k)
        for (int d = 0; d < N; ++d) o[d] += params_[idx[k] * N + d] * wt[k];
    }
  }
  return out;
}

// ---- importer (reference lib/Importer.cpp:25-38, 197-238) -------------------------------------------------
struct DepthVideoImporter {
  static void importVideo(DepthVideo& video, const std::string& path, bool discoverStreams) {
    std::ifstream is(path + "/frames.txt", std::ios::binary);
    if (is.fail()) throw std::runtime_error("Could not open frame file.");
    int n = -1, w = -1, h = -1;
    is >> n >> w >> h;
    if (n <= 0) throw std::runtime_error("Invalid frame file.");
    std::vector<float> pts(n);
    float minPts = 0.f;
    for (int i = 0; i < n; ++i) {
      float p;
      is >> p;
      if (i == 0) minPts = p;
      p -= minPts;
      if (i > 0 && p <= pts[i - 1]) throw std::runtime_error("Non-monotonic PTS detected.");
      pts[i] = p;
    }
    video.init(path, w, h, std::move(pts));
    if (discoverStreams) {
      const std::tuple<const char*, const char*, const char*, int> cs[] = {
          {"color_full", "full", ".png", 21}, {"color_down", "down", ".raw", 21},
          {"color_down_png", "down_png", ".png", 21}, {"dynamic_mask", "dynamic_mask", ".png", 0}};
      for (auto& c : cs)
        if (dirExists(path + "/" + std::get<0>(c)))
          video.createColorStream(std::get<1>(c), std::get<0>(c), std::get<2>(c), std::get<3>(c), {-1, -1});
    }
  }
};

// ---- flow constraints (reference lib/FlowConstraints.{h,cpp}) -----------------------------------------------
struct FlowConstraintsParams {
  FrameRange frameRange;
  int matchSeparation = 10;
  int minDynamicDistance = -1;
  bool doNotUseCache = false;
};

struct PairConstraints {
  std::vector<std::array<float, 4>> loc;
  std::vector<uint8_t> isStatic;
};
struct TripletConstraints {
  std::vector<std::array<float, 6>> loc;
  std::vector<uint8_t> isStatic;
};

struct FlowConstraintsCollection {
  static constexpr uint32_t kFileFormatVersion = 3;
  const DepthVideo* video_;
  std::string path_;
  FlowConstraintsParams params_;
  std::map<std::pair<int, int>, PairConstraints> pairs_;
  std::map<int, TripletConstraints> triplets_;
  // matchSeparation = 0 (reference :315-329, 381-395: every masked in-bounds pixel is a constraint): the flow fields and masks
  // the constraints were computed from are kept, in pairs_ order, so that DepthVideoProcessor can hand the SOLVER the images
  // (cvd_set_pair_flows: dense mode, 17 B per pixel pair) instead of the materialised list.  Valid while every constraint is
  // static and the collection is what compute() produced.
  std::vector<float> denseFlow_;
  std::vector<uint8_t> denseMask_;
  int denseW_ = 0, denseH_ = 0;
  bool denseValid_ = false;

  FlowConstraintsCollection(const DepthVideo& video, const FlowConstraintsParams& params)
      : video_(&video), path_(video.path_), params_(params) {  // reference :44-94
    const std::string listFile = path_ + "/flow_list.json";
    std::ifstream is(listFile);
    if (!is) throw std::runtime_error("Flow list file does not exist.");
    std::stringstream ss;
    ss << is.rdbuf();
    const std::string txt = ss.str();
    // JSON list of 2-element rows; row 0 is a header and is skipped (reference :59, written by flow.py:53)
    std::vector<std::pair<long, long>> rows;
    int depth = 0;
    std::vector<long> cur;
    std::string num;
    auto flush = [&]() { if (!num.empty()) { try { cur.push_back(std::stol(num)); } catch (...) { cur.push_back(-1); } num.clear(); } };
    bool inStr = false;
    for (char c : txt) {
      if (c == '"') { inStr = !inStr; continue; }
      if (inStr) continue;
      if (c == '[') { ++depth; cur.clear(); num.clear(); }
      else if (c == ']') { flush(); if (depth == 2) rows.push_back({cur.size() > 0 ? cur[0] : -1, cur.size() > 1 ? cur[1] : -1}); --depth; }
      else if (c == ',') flush();
      else if ((c >= '0' && c <= '9') || c == '-') num.push_back(c);
    }
    for (size_t i = 1; i < rows.size(); ++i) {
      const int a = static_cast<int>(rows[i].first), b = static_cast<int>(rows[i].second);
      if (!params.frameRange.inRange(a) || !params.frameRange.inRange(b)) continue;
      pairs_.emplace(std::make_pair(a, b), PairConstraints());
    }
    for (int t = params.frameRange.firstFrame() + 1; t <= params.frameRange.lastFrame() - 1; ++t) {
      if (!params.frameRange.inRange(t - 1) || !params.frameRange.inRange(t) || !params.frameRange.inRange(t + 1)) continue;
      triplets_.emplace(t, TripletConstraints());
    }
    if (params.doNotUseCache) {
      compute();
    } else if (!load()) {
      compute();
      save();
    }
  }

  // ---- image files of the dataset (reference lib/core/CvUtil.cpp:25-36 raw header; PNG through Pillow: no OpenCV) ----
  static std::vector<float> readRawFloat(const std::string& fn, int channels, int& rows, int& cols) {
    std::ifstream is(fn, std::ios::binary);
    if (!is) throw std::runtime_error("Could not open '" + fn + "'.");
    rows = rd<int32_t>(is);
    cols = rd<int32_t>(is);
    const int type = rd<int32_t>(is);
    const size_t elem = rd<size_t>(is);
    // CV_32FC(n) = 5 + 8 (n - 1)
    if (type != 5 + 8 * (channels - 1) || elem != static_cast<size_t>(4 * channels) || rows <= 0 || cols <= 0)
      throw std::runtime_error("Unexpected image type in '" + fn + "'.");
    std::vector<float> v(static_cast<size_t>(rows) * cols * channels);
    is.read(reinterpret_cast<char*>(v.data()), v.size() * 4);
    if (!is) throw std::runtime_error("Truncated image file '" + fn + "'.");
    return v;
  }
  static std::vector<uint8_t> readPngGray(const std::string& fn, int& rows, int& cols) {  // imread(IMREAD_GRAYSCALE)
    py::gil_scoped_acquire gil;
    py::object im = py::module_::import("PIL.Image").attr("open")(fn).attr("convert")("L");
    cols = im.attr("width").cast<int>();
    rows = im.attr("height").cast<int>();
    const std::string bytes = im.attr("tobytes")().cast<std::string>();
    if (bytes.size() != static_cast<size_t>(rows) * cols) throw std::runtime_error("Unexpected PNG payload in '" + fn + "'.");
    return std::vector<uint8_t>(bytes.begin(), bytes.end());
  }
  struct Device {
    cvd_handle* h = nullptr;
    explicit Device(int dev) : h(cvd_create(dev)) {
      if (!h) throw std::runtime_error(std::string("cvd_create: ") + cvd_last_error(nullptr));
    }
    ~Device() { if (h) cvd_destroy(h); }
    void check(int rc) const { if (rc != 0) throw std::runtime_error(cvd_last_error(h)); }
  };
  int device_ = 0;  // extension: HIP device used by compute() / setStaticFlagFromDynamicMask()

  // dynamicDistance of every frame in `frames` (reference :257-286); false when there is no dynamic_mask stream
  bool dynamicDistances(Device& dev, const std::vector<int>& frames, int numFrames, std::vector<float>& dist, int& dw,
                        int& dh) const {
    if (!video_->hasColorStream("dynamic_mask")) return false;
    const ColorStream& ms = *video_->colorStreams_.at(video_->colorStreamIndex("dynamic_mask"));
    std::vector<uint8_t> masks;
    dw = dh = 0;
    for (size_t k = 0; k < frames.size(); ++k) {
      int r, c;
      const std::string fn = ms.path_ + "/frame_" + fmtInt6(frames[k]) + ms.extension_;
      if (!fileExists(fn)) throw std::runtime_error("Dynamic mask stream is missing a frame.");
      std::vector<uint8_t> m = readPngGray(fn, r, c);
      if (k == 0) { dw = c; dh = r; masks.resize(static_cast<size_t>(frames.size()) * dw * dh); }
      if (c != dw || r != dh) throw std::runtime_error("Dynamic masks have inconsistent sizes.");
      std::copy(m.begin(), m.end(), masks.begin() + k * static_cast<size_t>(dw) * dh);
    }
    std::vector<float> packed(masks.size());
    dev.check(cvd_dynamic_distance(dev.h, static_cast<int>(frames.size()), dh, dw, masks.data(), packed.data(), nullptr));
    // the samplers index the maps by frame number
    dist.assign(static_cast<size_t>(numFrames) * dw * dh, 0.f);
    for (size_t k = 0; k < frames.size(); ++k)
      std::copy(packed.begin() + k * static_cast<size_t>(dw) * dh, packed.begin() + (k + 1) * static_cast<size_t>(dw) * dh,
                dist.begin() + static_cast<size_t>(frames[k]) * dw * dh);
    return true;
  }

  // FlowConstraintsCollection::compute (reference :288-550): corner response of every frame, then the greedy disk
  // sampling of every pair / triplet, all on the device (cvd_corner_min_eigenval, cvd_dynamic_distance,
  // cvd_sample_pair_constraints, cvd_sample_triplet_constraints); this function only moves files to buffers.
  void compute() {
    const ColorStream& cs = *video_->colorStreams_.at(video_->colorStreamIndex("down"));
    const int F = video_->numFrames();
    std::vector<int> frames;  // frames that appear in a key
    {
      std::vector<char> used(F, 0);
      for (auto& kv : pairs_) { used.at(kv.first.first) = 1; used.at(kv.first.second) = 1; }
      for (auto& kv : triplets_) { used.at(kv.first - 1) = 1; used.at(kv.first) = 1; used.at(kv.first + 1) = 1; }
      for (int f = 0; f < F; ++f) if (used[f]) frames.push_back(f);
    }
    if (frames.empty()) return;
    Device dev(device_);
    int w = 0, h = 0;
    std::vector<float> corner;
    {
      std::vector<float> bgr;
      for (size_t k = 0; k < frames.size(); ++k) {
        int r, c;
        std::vector<float> im = readRawFloat(cs.path_ + "/frame_" + fmtInt6(frames[k]) + cs.extension_, 3, r, c);
        if (k == 0) { w = c; h = r; bgr.resize(frames.size() * static_cast<size_t>(w) * h * 3); }
        if (c != w || r != h) throw std::runtime_error("Color frames have inconsistent sizes.");
        std::copy(im.begin(), im.end(), bgr.begin() + k * static_cast<size_t>(w) * h * 3);
      }
      std::vector<float> packed(frames.size() * static_cast<size_t>(w) * h);
      dev.check(cvd_corner_min_eigenval(dev.h, static_cast<int>(frames.size()), h, w, bgr.data(), packed.data(), nullptr));
      corner.assign(static_cast<size_t>(F) * w * h, 0.f);
      for (size_t k = 0; k < frames.size(); ++k)
        std::copy(packed.begin() + k * static_cast<size_t>(w) * h, packed.begin() + (k + 1) * static_cast<size_t>(w) * h,
                  corner.begin() + static_cast<size_t>(frames[k]) * w * h);
    }
    dev.check(cvd_set_video(dev.h, F, w, h, video_->aspect_, video_->invAspect_));
    std::vector<float> dyn;
    int dw = 0, dh = 0;
    const bool haveDyn = dynamicDistances(dev, frames, F, dyn, dw, dh);
    auto loadFlowAndMask = [&](int a, int b, float* flow, uint8_t* mask) {  // reference :226-255
      const std::string ff = path_ + "/flow/flow_" + fmtInt6(a) + "_" + fmtInt6(b) + ".raw";
      if (!fileExists(ff)) throw std::runtime_error("Flow file does not exist.");
      int r, c;
      std::vector<float> fl = readRawFloat(ff, 2, r, c);
      if (c != w || r != h) throw std::runtime_error("Flow has the wrong size.");
      std::copy(fl.begin(), fl.end(), flow);
      const std::string mf = path_ + "/flow_mask/mask_" + fmtInt6(a) + "_" + fmtInt6(b) + ".png";
      if (!fileExists(mf)) throw std::runtime_error("Mask file does not exist.");
      std::vector<uint8_t> m = readPngGray(mf, r, c);
      if (c != w || r != h) throw std::runtime_error("Mask has the wrong size.");
      std::copy(m.begin(), m.end(), mask);
    };
    const size_t px = static_cast<size_t>(w) * h;
    const size_t batch = std::max<size_t>(1, (512ull << 20) / (px * 18));  // ~512 MiB of flow + mask per call
    // (with a dynamic mask the sampler rejects pixels whose distance is not > minDynamicDistance -- the pixels INSIDE the mask
    // (distance 0) already at minDynamicDistance = 0 --: the kept images would then hold constraints the list does not)
    const bool keepDense = params_.matchSeparation == 0 && !(haveDyn && params_.minDynamicDistance >= 0);
    denseFlow_.clear();
    denseMask_.clear();
    denseValid_ = false;
    {  // pairs
      std::vector<std::pair<int, int>> keys;
      for (auto& kv : pairs_) keys.push_back(kv.first);
      for (size_t k0 = 0; k0 < keys.size(); k0 += batch) {
        const size_t n = std::min(batch, keys.size() - k0);
        std::vector<int32_t> pf(2 * n);
        std::vector<float> flow(n * px * 2);
        std::vector<uint8_t> mask(n * px);
        for (size_t k = 0; k < n; ++k) {
          pf[2 * k] = keys[k0 + k].first;
          pf[2 * k + 1] = keys[k0 + k].second;
          loadFlowAndMask(pf[2 * k], pf[2 * k + 1], flow.data() + k * px * 2, mask.data() + k * px);
        }
        std::vector<int64_t> off(n + 1, 0);
        dev.check(cvd_sample_pair_constraints(dev.h, static_cast<int>(n), pf.data(), corner.data(), flow.data(), mask.data(),
                                              haveDyn ? dyn.data() : nullptr, dw, dh, params_.matchSeparation,
                                              static_cast<float>(params_.minDynamicDistance), off.data()));
        std::vector<float> loc(static_cast<size_t>(off[n]) * 4);
        if (!loc.empty()) dev.check(cvd_get_sampled_constraints(dev.h, loc.data()));
        if (keepDense) {
          denseFlow_.insert(denseFlow_.end(), flow.begin(), flow.end());
          denseMask_.insert(denseMask_.end(), mask.begin(), mask.end());
        }
        for (size_t k = 0; k < n; ++k) {
          PairConstraints& pc = pairs_.at(keys[k0 + k]);
          const size_t cnt = static_cast<size_t>(off[k + 1] - off[k]);
          pc.loc.resize(cnt);
          std::memcpy(pc.loc.data(), loc.data() + static_cast<size_t>(off[k]) * 4, cnt * 16);
          pc.isStatic.assign(cnt, 1);
        }
      }
    }
    if (keepDense && !pairs_.empty()) {
      denseW_ = w;
      denseH_ = h;
      denseValid_ = true;
    }
    {  // triplets (reference :467-550): flows centre -> previous and centre -> next
      std::vector<int> keys;
      for (auto& kv : triplets_) keys.push_back(kv.first);
      const size_t tb = std::max<size_t>(1, batch / 2);
      for (size_t k0 = 0; k0 < keys.size(); k0 += tb) {
        const size_t n = std::min(tb, keys.size() - k0);
        std::vector<int32_t> ce(n);
        std::vector<float> f10(n * px * 2), f12(n * px * 2);
        std::vector<uint8_t> m10(n * px), m12(n * px);
        for (size_t k = 0; k < n; ++k) {
          ce[k] = keys[k0 + k];
          loadFlowAndMask(ce[k], ce[k] - 1, f10.data() + k * px * 2, m10.data() + k * px);
          loadFlowAndMask(ce[k], ce[k] + 1, f12.data() + k * px * 2, m12.data() + k * px);
        }
        std::vector<int64_t> off(n + 1, 0);
        dev.check(cvd_sample_triplet_constraints(dev.h, static_cast<int>(n), ce.data(), corner.data(), f10.data(), m10.data(),
                                                 f12.data(), m12.data(), haveDyn ? dyn.data() : nullptr, dw, dh,
                                                 params_.matchSeparation, static_cast<float>(params_.minDynamicDistance),
                                                 off.data()));
        std::vector<float> loc(static_cast<size_t>(off[n]) * 6);
        if (!loc.empty()) dev.check(cvd_get_sampled_triplet_constraints(dev.h, loc.data()));
        for (size_t k = 0; k < n; ++k) {
          TripletConstraints& tc = triplets_.at(keys[k0 + k]);
          const size_t cnt = static_cast<size_t>(off[k + 1] - off[k]);
          tc.loc.resize(cnt);
          std::memcpy(tc.loc.data(), loc.data() + static_cast<size_t>(off[k]) * 6, cnt * 24);
          tc.isStatic.assign(cnt, 1);
        }
      }
    }
  }

  bool load() {  // reference :116-189
    const std::string fn = path_ + "/flow_constraints.dat";
    if (!fileExists(fn)) return false;
    std::ifstream is(fn, std::ios::binary);
    if (rd<uint32_t>(is) != 0xDEADBEEF) throw std::runtime_error("Did not see magic marker at beginning of file.");
    const uint32_t fmt = rd<uint32_t>(is);
    if (fmt > kFileFormatVersion) throw std::runtime_error("File format too new.");
    if (fmt < 3) throw std::runtime_error("File format too old.");
    if (rd<int32_t>(is) != params_.matchSeparation) return false;
    for (auto& kv : pairs_) {
      const int a = rd<int32_t>(is), b = rd<int32_t>(is);
      if (a != kv.first.first || b != kv.first.second) throw std::runtime_error("Read incorrect pair from file.");
      const size_t n = rd<size_t>(is);
      kv.second.loc.resize(n);
      is.read(reinterpret_cast<char*>(kv.second.loc.data()), n * 16);
      kv.second.isStatic.assign(n, 1);  // isStatic is not serialised (reference lib/FlowConstraints.h:96-104)
    }
    for (auto& kv : triplets_) {
      if (rd<int32_t>(is) != kv.first) throw std::runtime_error("Read incorrect triplet from file.");
      const size_t n = rd<size_t>(is);
      kv.second.loc.resize(n);
      is.read(reinterpret_cast<char*>(kv.second.loc.data()), n * 24);
      kv.second.isStatic.assign(n, 1);
    }
    if (rd<uint32_t>(is) != 0xDEADBEEF) throw std::runtime_error("Did not see magic marker at end of file.");
    return true;
  }
  void save() {  // reference :191-224
    std::ofstream os(path_ + "/flow_constraints.dat", std::ios::binary);
    wr<uint32_t>(os, 0xDEADBEEF);
    wr<uint32_t>(os, kFileFormatVersion);
    wr<int32_t>(os, params_.matchSeparation);
    for (auto& kv : pairs_) {
      wr<int32_t>(os, kv.first.first); wr<int32_t>(os, kv.first.second);
      wr<size_t>(os, kv.second.loc.size());
      os.write(reinterpret_cast<const char*>(kv.second.loc.data()), kv.second.loc.size() * 16);
    }
    for (auto& kv : triplets_) {
      wr<int32_t>(os, kv.first);
      wr<size_t>(os, kv.second.loc.size());
      os.write(reinterpret_cast<const char*>(kv.second.loc.data()), kv.second.loc.size() * 24);
    }
    wr<uint32_t>(os, 0xDEADBEEF);
  }
  void resetStaticFlag() {
    for (auto& kv : pairs_) std::fill(kv.second.isStatic.begin(), kv.second.isStatic.end(), 1);
    for (auto& kv : triplets_) std::fill(kv.second.isStatic.begin(), kv.second.isStatic.end(), 1);
  }
  // dense hand-over is valid only while every pair constraint is static (the images carry no per-constraint flags)
  bool allPairConstraintsStatic() const {
    for (auto& kv : pairs_)
      for (uint8_t st : kv.second.isStatic)
        if (!st) return false;
    return true;
  }
  void setStaticFlagFromDynamicMask(int distance) {  // reference :573-660
    if (!video_->hasColorStream("dynamic_mask")) { resetStaticFlag(); return; }
    const int F = video_->numFrames();
    std::vector<int> frames;
    {
      std::vector<char> used(F, 0);
      for (auto& kv : pairs_) { used.at(kv.first.first) = 1; used.at(kv.first.second) = 1; }
      for (auto& kv : triplets_) { used.at(kv.first - 1) = 1; used.at(kv.first) = 1; used.at(kv.first + 1) = 1; }
      for (int f = 0; f < F; ++f) if (used[f]) frames.push_back(f);
    }
    if (frames.empty()) return;
    Device dev(device_);
    std::vector<float> dd;
    int w = 0, h = 0;
    dynamicDistances(dev, frames, F, dd, w, h);
    // static <=> every end point is farther than `distance` from the dynamic region; pixel = int(loc * w) for both
    // coordinates (loc.y is scaled by invAspect, so `* w` lands on the row: reference :616-619)
    auto isFar = [&](int frame, float lx, float ly) {
      const int ix = static_cast<int>(lx * w), iy = static_cast<int>(ly * w);
      if (ix < 0 || ix >= w || iy < 0 || iy >= h) throw std::runtime_error("constraint outside the dynamic mask");
      return dd[(static_cast<size_t>(frame) * h + iy) * w + ix] > static_cast<float>(distance);
    };
    for (auto& kv : pairs_)
      for (size_t i = 0; i < kv.second.loc.size(); ++i) {
        const auto& c = kv.second.loc[i];
        kv.second.isStatic[i] = isFar(kv.first.first, c[0], c[1]) && isFar(kv.first.second, c[2], c[3]);
      }
    for (auto& kv : triplets_)
      for (size_t i = 0; i < kv.second.loc.size(); ++i) {
        const auto& c = kv.second.loc[i];
        kv.second.isStatic[i] = isFar(kv.first - 1, c[0], c[1]) && isFar(kv.first, c[2], c[3]) && isFar(kv.first + 1, c[4], c[5]);
      }
  }
  // reference lib/FlowConstraints.cpp:662-748: every NON-static pair constraint stamps a disk of radius `distance` around
  // its end point into the mask of that end point's frame; afterwards every pair / triplet constraint with an end point on
  // a stamped pixel becomes non-static as well.  Pixel = int(loc * w) for both coordinates (:696-697: loc.y is scaled by
  // invAspect, so `* w` lands on the row); raster = the "down" colour stream.  The reference's std::map iteration order
  // is irrelevant: the masks are complete before the second pass.  Host loop over a host container, as in the reference.
  void pruneStaticFlag(int distance) {
    if (distance < 0) throw std::runtime_error("pruneStaticFlag: negative distance");
    if (!video_->hasColorStream("down")) throw std::runtime_error("Color stream 'down' does not exist.");
    const ColorStream& down = *video_->colorStreams_[video_->colorStreamIndex("down")];
    int w = down.width_, h = down.height_;
    if (w <= 0 || h <= 0) {
      // stream created without an explicit size: the reference's ColorStream takes it from its first image
      std::ifstream is(down.path_ + "/frame_" + fmtInt6(0) + down.extension_, std::ios::binary);
      if (is) {
        h = rd<int32_t>(is);
        w = rd<int32_t>(is);
      } else {  // no images on disk (constraints came from the cache file): the video's own raster
        w = video_->width_;
        h = video_->height_;
      }
    }
    if (w <= 0 || h <= 0) throw std::runtime_error("pruneStaticFlag: the 'down' colour stream has no size");
    const int F = video_->numFrames();
    std::vector<std::vector<uint8_t>> masks(F);
    auto maskOf = [&](int f) -> std::vector<uint8_t>& {
      std::vector<uint8_t>& m = masks.at(f);
      if (m.empty()) m.assign(static_cast<size_t>(w) * h, 0);
      return m;
    };
    const long long r2 = static_cast<long long>(distance) * distance;
    auto stamp = [&](int f, float lx, float ly) {
      const int x = static_cast<int>(lx * w), y = static_cast<int>(ly * w);
      std::vector<uint8_t>& m = maskOf(f);
      for (int my = std::max(0, y - distance); my <= std::min(h - 1, y + distance); ++my)
        for (int mx = std::max(0, x - distance); mx <= std::min(w - 1, x + distance); ++mx)
          if (static_cast<long long>(mx - x) * (mx - x) + static_cast<long long>(my - y) * (my - y) <= r2)  // buildDiskMask
            m[static_cast<size_t>(my) * w + mx] = 255;
    };
    for (auto& kv : pairs_)
      for (size_t i = 0; i < kv.second.loc.size(); ++i) {
        if (kv.second.isStatic[i]) continue;
        const auto& c = kv.second.loc[i];
        stamp(kv.first.first, c[0], c[1]);
        stamp(kv.first.second, c[2], c[3]);
      }
    auto hit = [&](int f, float lx, float ly) {
      const std::vector<uint8_t>& m = masks.at(f);
      if (m.empty()) return false;
      const int x = static_cast<int>(lx * w), y = static_cast<int>(ly * w);
      if (x < 0 || x >= w || y < 0 || y >= h) throw std::runtime_error("constraint outside the 'down' raster");
      return m[static_cast<size_t>(y) * w + x] != 0;
    };
    for (auto& kv : pairs_)
      for (size_t i = 0; i < kv.second.loc.size(); ++i) {
        const auto& c = kv.second.loc[i];
        if (hit(kv.first.first, c[0], c[1]) || hit(kv.first.second, c[2], c[3])) kv.second.isStatic[i] = 0;
      }
    for (auto& kv : triplets_)
      for (size_t i = 0; i < kv.second.loc.size(); ++i) {
        const auto& c = kv.second.loc[i];
        if (hit(kv.first - 1, c[0], c[1]) || hit(kv.first, c[2], c[3]) || hit(kv.first + 1, c[4], c[5])) kv.second.isStatic[i] = 0;
      }
  }
  // extension: explicit flags (what setStaticFlagFromDynamicMask would compute), per pair in map order
  void setStaticFlags(int a, int b, const std::vector<uint8_t>& flags) {
    auto& pc = pairs_.at({a, b});
    if (flags.size() != pc.isStatic.size()) throw std::runtime_error("flag count mismatch");
    pc.isStatic = flags;
  }
  int numPairs() const { return static_cast<int>(pairs_.size()); }
  long numConstraints() const { long n = 0; for (auto& kv : pairs_) n += static_cast<long>(kv.second.loc.size()); return n; }
};

// ---- optimizer params + processor ------------------------------------------------------------------------------
struct DvpoParams {  // reference lib/PoseOptimizer.h:54-108
  FrameRange frameRange;
  int maxIterations = 1000, numThreads = 12, numSteps = 4;
  double robustness = 0.5;
  bool huberLoss = false;  // extension (no reference counterpart): ceres::HuberLoss(robustness) instead of the CauchyLoss
  StaticLossType staticLossType = StaticLossType::ReproDisparity;
  double staticSpatialWeight = 1.0, staticDepthWeight = 1.0;
  SmoothLossType smoothLossType = SmoothLossType::ReproDisparityLaplacian;
  double smoothStaticWeight = 0.0, smoothDynamicWeight = 0.0;
  double positionReg = 0.0, scaleReg = 1.0;
  int scaleRegGridSize = 10;
  double depthDeformRegInitial = 1.0, depthDeformRegFinal = 0.1, adaptiveDeformationCost = 0.0, spatialDeformReg = 1.0;
  bool graduateDepthDeformReg = false;
  double focalReg = 1.0;
  bool coarseToFine = true;
  int ctfLong = 17, ctfShort = 10;
  bool deferredSpatialOpt = false;
  int dsoLong = 4, dsoShort = 3;
  double focalLong = 0.3461538376301239;
  IntrinsicsOptimization intrOpt = IntrinsicsOptimization::PerFrame;
  bool fixPoses = false, fixDepthXforms = false, fixSpatialXforms = false;
  bool normalizeDepthFromFirstFrame = true;
};

enum class Op { None, Reset, Copy, BilateralFilter, FlowGuidedFilter, ClipMaxDepth, ComputeConstraints,
                ResetConstraintStaticFlag, SetConstraintStaticFlagFromDynamicMask, PruneConstraintStaticFlag,
                ComputeTracks, GridXformSplit, ResetPoses, ResetDepthXforms, ResetSpatialXforms, NormalizeDepth,
                OptimizePoses, ResetNormalizeOptimize };

struct DvpParams {  // reference lib/Processor.h:60-90
  Op op = Op::None;
  FrameRange frameRange;
  int colorStream = 0, depthStream = 0, sourceDepthStream = 0, spatialRadius = 0, frameRadius = 2;
  float depthSigma = 0.3f, colorSigma = 0.0f;
  bool median = false, farConnections = false;
  float maxDepth = 1000.f;
  int matchSeparation = 10;
  float flowConsistancyThresh = 0.05f;
  int trackSpawnDistance = 20, trackPruneDistance = 5, minDynamicDistance = 3, minTrackLength = 4;
  XformDescriptor depthXformDesc, spatialXformDesc;
  DvpoParams poseOptimizer;
};

struct DepthVideoProcessor {
  DepthVideo* video_;
  int device_ = 0;
  bool usedFlowImages_ = false;  // the last normalizeDepth / optimizePoses handed the solver flow images (dense mode)
  explicit DepthVideoProcessor(DepthVideo* v) : video_(v) {}

  void resetPoses(const DvpParams& p) {  // reference lib/Processor.cpp:987-1003
    DepthStream& ds = *video_->depthStreams_.at(p.depthStream);
    for (auto& f : ds.frames_) {
      f->extrinsics = Extrinsics();
      const float focal = static_cast<float>(p.poseOptimizer.focalLong);
      if (video_->aspect_ >= 1.f) {
        f->intrinsics.hFov = std::atan(focal) * 2.f;
        f->intrinsics.vFov = std::atan(focal / video_->aspect_) * 2.f;
      } else {
        f->intrinsics.hFov = std::atan(focal * video_->aspect_) * 2.f;
        f->intrinsics.vFov = std::atan(focal) * 2.f;
      }
    }
  }
  void resetDepthXforms(const DvpParams& p) { video_->depthStreams_.at(p.depthStream)->resetDepthXforms(p.depthXformDesc); }
  void resetSpatialXforms(const DvpParams& p) { video_->depthStreams_.at(p.depthStream)->resetSpatialXforms(p.spatialXformDesc); }
  void reset(const DvpParams& p) { for (int f : p.frameRange.frames) video_->depthStreams_.at(p.depthStream)->frame(f).clear(); }

  // ---- marshalling to / from the C ABI ---------------------------------------------------------------------
  struct Session {
    cvd_handle* h = nullptr;
    ~Session() { if (h) cvd_destroy(h); }
    void check(int rc) const { if (rc != 0) throw std::runtime_error(cvd_last_error(h)); }
  };
  static cvd_opt_params toC(const DvpoParams& p, std::vector<int32_t>& range) {
    cvd_opt_params c;
    cvd_opt_params_default(&c);
    range.assign(p.frameRange.frames.begin(), p.frameRange.frames.end());
    c.frame_range = range.empty() ? nullptr : range.data();
    c.num_range_frames = static_cast<int32_t>(range.size());
    c.max_iterations = p.maxIterations; c.num_threads = p.numThreads; c.num_steps = p.numSteps; c.robustness = p.robustness;
    c.static_loss_type = static_cast<int>(p.staticLossType);
    c.static_spatial_weight = p.staticSpatialWeight; c.static_depth_weight = p.staticDepthWeight;
    c.smooth_loss_type = static_cast<int>(p.smoothLossType);
    c.smooth_static_weight = p.smoothStaticWeight; c.smooth_dynamic_weight = p.smoothDynamicWeight;
    c.position_reg = p.positionReg; c.scale_reg = p.scaleReg; c.scale_reg_grid_size = p.scaleRegGridSize;
    c.depth_deform_reg_initial = p.depthDeformRegInitial; c.depth_deform_reg_final = p.depthDeformRegFinal;
    c.adaptive_deformation_cost = p.adaptiveDeformationCost; c.spatial_deform_reg = p.spatialDeformReg;
    c.graduate_depth_deform_reg = p.graduateDepthDeformReg; c.focal_reg = p.focalReg;
    c.coarse_to_fine = p.coarseToFine; c.ctf_long = p.ctfLong; c.ctf_short = p.ctfShort;
    c.deferred_spatial_opt = p.deferredSpatialOpt; c.dso_long = p.dsoLong; c.dso_short = p.dsoShort;
    c.focal_long = p.focalLong; c.intr_opt = static_cast<int>(p.intrOpt);
    c.fix_poses = p.fixPoses; c.fix_depth_xforms = p.fixDepthXforms; c.fix_spatial_xforms = p.fixSpatialXforms;
    c.normalize_depth_from_first_frame = p.normalizeDepthFromFirstFrame;
    return c;
  }
  void upload(Session& s, const DvpParams& p, const FlowConstraintsCollection& fc, bool forNormalize) {
    DepthStream& ds = *video_->depthStreams_.at(p.depthStream);
    s.h = cvd_create(device_);
    if (!s.h) throw std::runtime_error(cvd_last_error(nullptr));
    const int F = video_->numFrames();
    // every frame of the range needs its source depth (reference lib/PoseOptimizer.cpp:108-111)
    int w = -1, h = -1;
    for (int f = 0; f < F; ++f)
      if (ds.frame(f).sourceDepth()) { w = ds.width_; h = ds.height_; break; }
    if (w <= 0) throw std::runtime_error("Missing depth image.");
    s.check(cvd_set_video(s.h, F, w, h, video_->aspect_, video_->invAspect_));
    std::vector<cvd_frame_pose> poses(F);
    for (int f = 0; f < F; ++f) {
      DepthFrame& df = ds.frame(f);
      const std::vector<float>* d = df.sourceDepth();
      const bool need = p.poseOptimizer.frameRange.frames.empty() || p.poseOptimizer.frameRange.frames.count(f);
      if (!d && need) throw std::runtime_error("Missing depth image.");
      if (d) s.check(cvd_set_depth(s.h, f, d->data()));
      cvd_frame_pose& q = poses[f];
      for (int i = 0; i < 3; ++i) q.position[i] = df.extrinsics.position[i];
      q.orientation[0] = df.extrinsics.orientation.x_; q.orientation[1] = df.extrinsics.orientation.y_;
      q.orientation[2] = df.extrinsics.orientation.z_; q.orientation[3] = df.extrinsics.orientation.w_;
      q.vfov = df.intrinsics.vFov;
      q.hfov = df.intrinsics.hFov;
    }
    s.check(cvd_set_poses(s.h, poses.data()));
    if (p.poseOptimizer.adaptiveDeformationCost > 0.0 && video_->hasColorStream("dynamic_mask")) {
      // AdaptiveDeformationCost reads the frame's dynamic mask (reference lib/PoseOptimizer.cpp:1469-1481); without the
      // stream the solver raises the reference's "Adaptive smoothness requires a dynamic mask stream."
      const ColorStream& ms = *video_->colorStreams_.at(video_->colorStreamIndex("dynamic_mask"));
      std::vector<uint8_t> masks;
      int mw = 0, mh = 0;
      for (int f = 0; f < F; ++f) {
        const bool need = p.poseOptimizer.frameRange.frames.empty() || p.poseOptimizer.frameRange.frames.count(f);
        const std::string fn = ms.path_ + "/frame_" + fmtInt6(f) + ms.extension_;
        if (!fileExists(fn)) {
          if (need) throw std::runtime_error("Dynamic mask stream is missing a frame.");
          continue;
        }
        int r, c;
        const std::vector<uint8_t> m = FlowConstraintsCollection::readPngGray(fn, r, c);
        if (masks.empty()) { mw = c; mh = r; masks.assign(static_cast<size_t>(F) * mw * mh, 255); }
        if (c != mw || r != mh) throw std::runtime_error("Dynamic masks have inconsistent sizes.");
        std::copy(m.begin(), m.end(), masks.begin() + static_cast<size_t>(f) * mw * mh);
      }
      if (!masks.empty()) s.check(cvd_set_dynamic_masks(s.h, mh, mw, masks.data()));
    }
    // matchSeparation = 0 collections whose flow images are at hand and match the depth stream's raster go to the solver as
    // images (dense mode: nothing materialised on the device); lib_python.setDenseHandOver(False) forces the list (comparison / tests)
    // -- and only when the solve these parameters describe lies within the dense mode's scope (the library fails outside it
    // instead of falling back: the list path serves every configuration)
    std::vector<int32_t> scopeRange;
    const cvd_opt_params scopeParams = toC(p.poseOptimizer, scopeRange);
    const cvd_xform_desc scopeDd = ds.depthXformDesc_.toC(), scopeSd = ds.spatialXformDesc_.toC();
    const bool denseHandOver = fc.denseValid_ && fc.denseW_ == w && fc.denseH_ == h && g_denseHandOver &&
                               fc.denseMask_.size() == fc.pairs_.size() * static_cast<size_t>(w) * h && fc.allPairConstraintsStatic() &&
                               cvd_dense_mode_supported(&scopeParams, &scopeDd, &scopeSd, fc.triplets_.empty() ? 0 : 1, 1,
                                                        forNormalize ? 1 : 0) == 1;
    usedFlowImages_ = denseHandOver;
    if (denseHandOver) {
      std::vector<int32_t> pf;
      for (auto& kv : fc.pairs_) {
        pf.push_back(kv.first.first);
        pf.push_back(kv.first.second);
      }
      s.check(cvd_set_pair_flows(s.h, static_cast<int>(pf.size() / 2), pf.data(), fc.denseFlow_.data(), fc.denseMask_.data()));
    } else {
      std::vector<int32_t> pf;
      std::vector<int64_t> off{0};
      std::vector<float> loc;
      std::vector<uint8_t> st;
      for (auto& kv : fc.pairs_) {
        pf.push_back(kv.first.first);
        pf.push_back(kv.first.second);
        for (size_t i = 0; i < kv.second.loc.size(); ++i) {
          loc.insert(loc.end(), kv.second.loc[i].begin(), kv.second.loc[i].end());
          st.push_back(kv.second.isStatic[i]);
        }
        off.push_back(static_cast<int64_t>(st.size()));
      }
      s.check(cvd_set_pair_constraints(s.h, static_cast<int>(pf.size() / 2), pf.data(), off.data(), loc.data(), st.data()));
    }
    if (!fc.triplets_.empty()) {
      // scene-flow smoothness triplets, keyed by the centre frame (reference lib/FlowConstraints.h:156-167); groups at
      // the sequence ends cannot form a triple and are not handed over
      std::vector<int32_t> centers;
      std::vector<int64_t> toff{0};
      std::vector<float> tloc;
      std::vector<uint8_t> tst;
      for (auto& kv : fc.triplets_) {
        if (kv.first < 1 || kv.first + 1 >= F) continue;
        centers.push_back(kv.first);
        for (size_t i = 0; i < kv.second.loc.size(); ++i) {
          tloc.insert(tloc.end(), kv.second.loc[i].begin(), kv.second.loc[i].end());
          tst.push_back(kv.second.isStatic[i]);
        }
        toff.push_back(static_cast<int64_t>(tst.size()));
      }
      if (!centers.empty())
        s.check(cvd_set_triplet_constraints(s.h, static_cast<int>(centers.size()), centers.data(), toff.data(), tloc.data(),
                                            tst.data()));
    }
    const cvd_xform_desc dd = ds.depthXformDesc_.toC(), sd = ds.spatialXformDesc_.toC();
    s.check(cvd_reset_depth_xforms(s.h, &dd));
    s.check(cvd_reset_spatial_xforms(s.h, &sd));
    for (int spatial = 0; spatial < 2; ++spatial) {
      const int np = cvd_num_xform_params(s.h, spatial);
      if (np <= 0) continue;
      std::vector<double> all(static_cast<size_t>(F) * np);
      for (int f = 0; f < F; ++f) {
        const Xform& x = spatial ? static_cast<const Xform&>(ds.frame(f).spatialXform()) : ds.frame(f).depthXform();
        if (x.numParams() != np) throw std::runtime_error("Transform parameter count mismatch.");
        std::copy(x.params_.begin(), x.params_.end(), all.begin() + static_cast<size_t>(f) * np);
      }
      s.check(cvd_set_xform_params(s.h, spatial, all.data()));
    }
  }
  void download(Session& s, const DvpParams& p, bool posesToo) {
    DepthStream& ds = *video_->depthStreams_.at(p.depthStream);
    const int F = video_->numFrames();
    cvd_xform_desc dd, sd;
    s.check(cvd_get_xform_desc(s.h, 0, &dd));
    s.check(cvd_get_xform_desc(s.h, 1, &sd));
    const XformDescriptor nd = XformDescriptor::fromC(dd), ns = XformDescriptor::fromC(sd);
    if (!(nd == ds.depthXformDesc_)) ds.resetDepthXforms(nd);       // coarse-to-fine changed the grid
    if (!(ns == ds.spatialXformDesc_)) ds.resetSpatialXforms(ns);
    for (int spatial = 0; spatial < 2; ++spatial) {
      const int np = cvd_num_xform_params(s.h, spatial);
      if (np <= 0) continue;
      std::vector<double> all(static_cast<size_t>(F) * np);
      s.check(cvd_get_xform_params(s.h, spatial, all.data()));
      for (int f = 0; f < F; ++f) {
        Xform& x = spatial ? static_cast<Xform&>(ds.frame(f).spatialXform()) : ds.frame(f).depthXform();
        x.params_.assign(all.begin() + static_cast<size_t>(f) * np, all.begin() + static_cast<size_t>(f + 1) * np);
      }
    }
    if (posesToo) {
      std::vector<cvd_frame_pose> poses(F);
      s.check(cvd_get_poses(s.h, poses.data()));
      for (int f = 0; f < F; ++f) {
        DepthFrame& df = ds.frame(f);
        for (int i = 0; i < 3; ++i) df.extrinsics.position[i] = poses[f].position[i];
        df.extrinsics.orientation.x_ = poses[f].orientation[0]; df.extrinsics.orientation.y_ = poses[f].orientation[1];
        df.extrinsics.orientation.z_ = poses[f].orientation[2]; df.extrinsics.orientation.w_ = poses[f].orientation[3];
        df.intrinsics.vFov = poses[f].vfov;
        df.intrinsics.hFov = poses[f].hfov;
      }
    }
  }
  void gridXformSplit(const DvpParams& p) {  // reference lib/Processor.cpp:888-985, through the library
    Session s;
    s.h = cvd_create(device_);
    if (!s.h) throw std::runtime_error(cvd_last_error(nullptr));
    DepthStream& ds = *video_->depthStreams_.at(p.depthStream);
    const int F = video_->numFrames();
    s.check(cvd_set_video(s.h, F, 2, 2, video_->aspect_, video_->invAspect_));
    const cvd_xform_desc dd = ds.depthXformDesc_.toC();
    s.check(cvd_reset_depth_xforms(s.h, &dd));
    const int np = cvd_num_xform_params(s.h, 0);
    std::vector<double> all(static_cast<size_t>(F) * std::max(np, 1));
    for (int f = 0; f < F; ++f) std::copy(ds.frame(f).depthXform().params_.begin(), ds.frame(f).depthXform().params_.end(), all.begin() + static_cast<size_t>(f) * np);
    if (np > 0) s.check(cvd_set_xform_params(s.h, 0, all.data()));
    const cvd_xform_desc nd = p.depthXformDesc.toC();
    s.check(cvd_grid_xform_split(s.h, &nd));
    ds.resetDepthXforms(p.depthXformDesc);
    const int nn = cvd_num_xform_params(s.h, 0);
    std::vector<double> out(static_cast<size_t>(F) * nn);
    s.check(cvd_get_xform_params(s.h, 0, out.data()));
    for (int f = 0; f < F; ++f) ds.frame(f).depthXform().params_.assign(out.begin() + static_cast<size_t>(f) * nn, out.begin() + static_cast<size_t>(f + 1) * nn);
  }
  void normalizeDepth(const DvpParams& p, const FlowConstraintsCollection& fc) {  // reference :1015-1019
    Session s;
    upload(s, p, fc, true);
    std::vector<int32_t> range;
    const cvd_opt_params c = toC(p.poseOptimizer, range);
    {
      py::gil_scoped_release nogil;
      s.check(cvd_normalize_depth(s.h, &c));
    }
    download(s, p, false);
  }
  void optimizePoses(const DvpParams& p, const FlowConstraintsCollection& fc) {  // reference :1021-1025
    Session s;
    upload(s, p, fc, false);
    std::vector<int32_t> range;
    const cvd_opt_params c = toC(p.poseOptimizer, range);
    if (p.poseOptimizer.huberLoss) {
      cvd_solver_options so;
      cvd_solver_options_default(&so);
      so.robust_loss = 1;
      s.check(cvd_set_solver_options(s.h, &so));
    }
    {
      py::gil_scoped_release nogil;
      s.check(cvd_pose_optimization(s.h, &c));
    }
    download(s, p, true);
    cvd_solve_summary sum;
    if (cvd_get_summary(s.h, &sum) == 0) {
      char b[256];
      std::snprintf(b, sizeof(b), "MI355X solve: %d LM iterations, %d PCG iterations, cost %.6e -> %.6e, %.3f s",
                    sum.num_iterations, sum.total_linear_iterations, sum.initial_cost, sum.final_cost, sum.total_seconds);
      logInfo(b);
    }
  }
  // Op::Copy, reference lib/Processor.cpp:152-181
  void copy(const DvpParams& p) {
    if (p.sourceDepthStream < 0 || p.sourceDepthStream >= video_->numDepthStreams())
      throw std::runtime_error("Source depth stream out of range.");
    if (p.sourceDepthStream == p.depthStream)
      throw std::runtime_error("Source and destination depth stream cannot be identical.");
    DepthStream& src = *video_->depthStreams_.at(p.sourceDepthStream);
    DepthStream& dst = *video_->depthStreams_.at(p.depthStream);
    for (int f : p.frameRange.frames) {
      DepthFrame& sf = src.frame(f);
      const std::vector<float>* d = sf.sourceDepth();
      if (!d) throw std::runtime_error("Source depth frame is missing.");
      const std::vector<float> x = sf.depthXform().apply(*d, sf.width(), sf.height());  // DepthFrame::depth()
      DepthFrame& df = dst.frame(f);
      checkDims(dst, sf.width(), sf.height());
      df.sourceDepth_ = x;
      df.triedLoad = true;
      df.intrinsics = sf.intrinsics;
      df.extrinsics = sf.extrinsics;
    }
  }

  // Op::FlowGuidedFilter, reference lib/Processor.cpp:315-590: gathers the files into one consecutive batch and runs
  // cvd_flow_guided_filter (robust_cvd_amd/csrc/cvd_filter.h) on it.
  void flowGuidedFilter(const DvpParams& p) {
    if (!p.frameRange.isConsecutive()) throw std::runtime_error("Frame range must be consecutive.");
    if (p.farConnections) throw std::runtime_error("flowGuidedFilter: farConnections is not supported by this build.");
    const int first = p.frameRange.firstFrame(), last = p.frameRange.lastFrame();
    const int lo = std::max(0, first - p.frameRadius);
    const int n = last - lo + 1;
    DepthStream& src = *video_->depthStreams_.at(p.sourceDepthStream);
    DepthStream& dst = *video_->depthStreams_.at(p.depthStream);
    int w = 0, h = 0;
    std::vector<float> ff, fb;
    std::vector<uint8_t> mf, mb;
    for (int k = 0; k + 1 < n && p.frameRadius > 0; ++k) {
      for (int dir = 0; dir < 2; ++dir) {
        const int a = lo + k + dir, b = lo + k + 1 - dir;  // forward: k -> k+1, backward: k+1 -> k
        int r, c;
        const std::string stem = fmtInt6(a) + "_" + fmtInt6(b);
        const std::vector<float> fl = FlowConstraintsCollection::readRawFloat(video_->path_ + "/flow/flow_" + stem + ".raw", 2, r, c);
        if (w == 0) {
          w = c; h = r;
          const size_t links = static_cast<size_t>(n - 1) * w * h;
          ff.resize(links * 2); fb.resize(links * 2); mf.resize(links); mb.resize(links);
        }
        if (c != w || r != h) throw std::runtime_error("Flow has the wrong size.");
        std::copy(fl.begin(), fl.end(), (dir ? fb : ff).begin() + static_cast<size_t>(k) * w * h * 2);
        const std::vector<uint8_t> m = FlowConstraintsCollection::readPngGray(video_->path_ + "/flow_mask/mask_" + stem + ".png", r, c);
        if (c != w || r != h) throw std::runtime_error("Mask has the wrong size.");
        std::copy(m.begin(), m.end(), (dir ? mb : mf).begin() + static_cast<size_t>(k) * w * h);
      }
    }
    const int dw = src.width(), dh = src.height();
    if (dw <= 0 || dh <= 0) throw std::runtime_error("Source depth stream has no frames.");
    if (w == 0) { w = dw; h = dh; }
    std::vector<float> depth(static_cast<size_t>(n) * dw * dh), cams(static_cast<size_t>(n) * 9);
    for (int k = 0; k < n; ++k) {
      DepthFrame& sf = src.frame(lo + k);
      const std::vector<float>* d = sf.sourceDepth();
      if (!d) throw std::runtime_error("Source depth frame is missing.");
      const std::vector<float> x = sf.depthXform().apply(*d, dw, dh);
      std::copy(x.begin(), x.end(), depth.begin() + static_cast<size_t>(k) * dw * dh);
      float* c = cams.data() + static_cast<size_t>(k) * 9;
      for (int i = 0; i < 3; ++i) c[i] = sf.extrinsics.position[i];
      c[3] = sf.extrinsics.orientation.x_; c[4] = sf.extrinsics.orientation.y_; c[5] = sf.extrinsics.orientation.z_;
      c[6] = sf.extrinsics.orientation.w_;
      c[7] = sf.intrinsics.hFov; c[8] = sf.intrinsics.vFov;
    }
    const int count = last - first + 1;
    std::vector<float> out(static_cast<size_t>(count) * w * h);
    Session s;
    s.h = cvd_create(device_);
    if (!s.h) throw std::runtime_error(std::string("cvd_create: ") + cvd_last_error(nullptr));
    s.check(cvd_flow_guided_filter(s.h, n, first - lo, count, h, w, dh, dw, video_->invAspect_, depth.data(), cams.data(),
                                   ff.data(), mf.data(), fb.data(), mb.data(), p.frameRadius, p.spatialRadius, p.median ? 1 : 0,
                                   out.data(), nullptr));
    for (int k = 0; k < count; ++k) {  // dstDs.frame(frame).setDepth(filteredDepth), reference :585
      DepthFrame& df = dst.frame(first + k);
      checkDims(dst, w, h);
      df.sourceDepth_.assign(out.begin() + static_cast<size_t>(k) * w * h, out.begin() + static_cast<size_t>(k + 1) * w * h);
      df.triedLoad = true;
    }
  }

  // Op::ClipMaxDepth, reference lib/Processor.cpp:592-617
  void clipMaxDepth(const DvpParams& p) {
    DepthStream& ds = *video_->depthStreams_.at(p.depthStream);
    for (int f : p.frameRange.frames) {
      DepthFrame& df = ds.frame(f);
      if (!df.enabled) continue;
      const std::vector<float>* d = df.sourceDepth();
      if (!d) continue;
      std::vector<float> x = df.depthXform().apply(*d, df.width(), df.height());  // DepthFrame::depth()
      for (float& v : x) v = std::min(v, p.maxDepth);
      df.sourceDepth_ = std::move(x);  // setDepth(clipped)
      df.triedLoad = true;
    }
  }

  void process(const DvpParams& p) {  // reference lib/Processor.cpp:115-144
    switch (p.op) {
      case Op::None: break;
      case Op::Reset: reset(p); break;
      case Op::Copy: copy(p); break;
      case Op::FlowGuidedFilter: flowGuidedFilter(p); break;
      case Op::ClipMaxDepth: clipMaxDepth(p); break;
      case Op::GridXformSplit: gridXformSplit(p); break;
      case Op::ResetPoses: resetPoses(p); break;
      case Op::ResetDepthXforms: resetDepthXforms(p); break;
      case Op::ResetSpatialXforms: resetSpatialXforms(p); break;
      default: throw std::runtime_error("Unsupported operation selected.");
    }
  }
};

}  // namespace cvdhost

using namespace cvdhost;

// numpy view of a std::array member that keeps its parent object alive (element assignment writes through)
template <typename T, size_t N>
static py::array_t<T> memberView(py::object parent, std::array<T, N>& member) {
  return py::array_t<T>({static_cast<py::ssize_t>(N)}, {static_cast<py::ssize_t>(sizeof(T))}, member.data(), parent);
}

PYBIND11_MODULE(lib_python, m) {
  m.doc() = "MI355X-native drop-in for robust_cvd's lib_python (optimizer path only)";
  m.def("initLib", []() {});
  m.def("logToStdout", []() { g_logStdout = true; });
  // (not in the reference: matchSeparation = 0 collections go to the solver as flow images when the solve lies in the dense
  // mode's scope; False keeps the materialised constraint list)
  m.def("setDenseHandOver", [](bool on) { g_denseHandOver = on; }, py::arg("enabled"));
  m.def("computeDepthRange", [](const py::array_t<float, py::array::c_style | py::array::forcecast>& d) {
    float mn = std::numeric_limits<float>::max(), mx = std::numeric_limits<float>::min();  // reference lib/DepthMapTransform.cpp:20-34
    for (ssize_t i = 0; i < d.size(); ++i) {
      const float v = d.data()[i];
      if (std::isfinite(v) && v > 0) { mn = std::min(mn, v); mx = std::max(mx, v); }
    }
    return std::make_tuple(mn, mx);
  });

  py::class_<Quaternionf>(m, "Quaternionf")
      .def(py::init())
      .def("x", [](Quaternionf& q) { return q.x_; }).def("y", [](Quaternionf& q) { return q.y_; })
      .def("z", [](Quaternionf& q) { return q.z_; }).def("w", [](Quaternionf& q) { return q.w_; })
      .def("coeffs", [](Quaternionf& q) { return std::array<float, 4>{q.x_, q.y_, q.z_, q.w_}; })
      .def("setCoeffs", [](Quaternionf& q, const std::array<float, 4>& c) { q.x_ = c[0]; q.y_ = c[1]; q.z_ = c[2]; q.w_ = c[3]; });
  py::class_<Extrinsics>(m, "Extrinsics")
      .def(py::init())
      // (the reference returns a numpy view of the Eigen member: `e.position[0] = x` writes through, lib/PythonBindings.cpp:181)
      .def_property("position", [](py::object self) { return memberView(self, self.cast<Extrinsics&>().position); },
                    [](Extrinsics& e, const std::array<float, 3>& v) { e.position = v; })
      .def_readwrite("orientation", &Extrinsics::orientation)
      .def("left", &Extrinsics::left).def("right", &Extrinsics::right).def("down", &Extrinsics::down)
      .def("up", &Extrinsics::up).def("forward", &Extrinsics::forward).def("backward", &Extrinsics::backward);
  py::class_<Intrinsics>(m, "Intrinsics")
      .def(py::init())
      .def_readwrite("vFov", &Intrinsics::vFov).def_readwrite("hFov", &Intrinsics::hFov)
      .def_readwrite("centerLat", &Intrinsics::centerLat).def_readwrite("centerLon", &Intrinsics::centerLon);

  py::enum_<ValueXformType>(m, "ValueXformType").value("None", ValueXformType::None).value("Scale", ValueXformType::Scale).value("ScaleShift", ValueXformType::ScaleShift);
  py::enum_<XformType>(m, "XformType").value("Depth", XformType::Depth).value("Spatial", XformType::Spatial);
  py::enum_<DepthXformType>(m, "DepthXformType").value("None", DepthXformType::None).value("Identity", DepthXformType::Identity)
      .value("Global", DepthXformType::Global).value("Grid", DepthXformType::Grid);
  py::enum_<SpatialXformType>(m, "SpatialXformType").value("None", SpatialXformType::None).value("Identity", SpatialXformType::Identity)
      .value("VerticalLinear", SpatialXformType::VerticalLinear).value("CornersBilinear", SpatialXformType::CornersBilinear)
      .value("BilinearGrid", SpatialXformType::BilinearGrid).value("BicubicGrid", SpatialXformType::BicubicGrid);

  py::class_<XformDescriptor>(m, "XformDescriptor")
      .def(py::init())
      .def_readwrite("type", &XformDescriptor::type).def_readwrite("depthType", &XformDescriptor::depthType)
      .def_readwrite("spatialType", &XformDescriptor::spatialType).def_readwrite("valueXform", &XformDescriptor::valueXform)
      // (numpy views of the members, as pybind11/eigen.h gives the reference: `d.gridSize[0] = 17` writes through,
      // lib/PythonBindings.cpp:237-238)
      .def_property("gridSize", [](py::object self) { return memberView(self, self.cast<XformDescriptor&>().gridSize); },
                    [](XformDescriptor& d, const std::array<int, 3>& v) { d.gridSize = v; })
      .def_property("depthMinMax", [](py::object self) { return memberView(self, self.cast<XformDescriptor&>().depthMinMax); },
                    [](XformDescriptor& d, const std::array<double, 2>& v) { d.depthMinMax = v; })
      .def_readwrite("cubicInterpolation", &XformDescriptor::cubicInterpolation)
      .def("reset", &XformDescriptor::reset, py::arg("type") = XformType::Depth)
      .def("str", &XformDescriptor::str).def("parse", &XformDescriptor::parse);

  py::class_<Xform>(m, "Xform")
      // Xform::clone (reference lib/DepthMapTransform.cpp:353-357, bound at lib/PythonBindings.cpp:246): a new transform of the
      // same descriptor with the parameters copied; Python owns the clone
      .def("clone", [](const Xform& x) -> py::object {
        if (x.desc_.type == XformType::Depth) {
          std::unique_ptr<DepthXform> c = createDepthXform(x.desc_);
          c->params_ = x.params_;
          return py::cast(std::move(c));
        }
        std::unique_ptr<SpatialXform> c = createSpatialXform(x.desc_);
        c->params_ = x.params_;
        return py::cast(std::move(c));
      })
      .def("copyFrom", &Xform::copyFrom)
      .def("desc", &Xform::desc, py::return_value_policy::copy)
      .def("str", &Xform::str)
      .def("params", &Xform::params, py::return_value_policy::copy)
      .def("setParams", [](Xform& x, const std::vector<double>& p) {
        if (static_cast<int>(p.size()) != x.numParams()) throw std::runtime_error("parameter count mismatch");
        x.params_ = p;
      })
      .def("numParams", &Xform::numParams);
  py::class_<DepthXform, Xform>(m, "DepthXform").def("paramMap", &DepthXform::paramMap);
  py::class_<SpatialXform, Xform>(m, "SpatialXform").def("warp", &SpatialXform::warp);

  py::class_<ColorStream>(m, "ColorStream")
      .def("name", [](ColorStream& c) { return c.name_; }).def("path", [](ColorStream& c) { return c.path_; })
      .def("extension", [](ColorStream& c) { return c.extension_; })
      .def("width", [](ColorStream& c) { return c.width_; }).def("height", [](ColorStream& c) { return c.height_; })
      .def("setDir", &ColorStream::setDir);

  py::class_<DepthFrame>(m, "DepthFrame")
      .def("sourceDepth", [](DepthFrame& f) -> py::object {
        const std::vector<float>* d = f.sourceDepth();
        if (!d) return py::none();
        py::array_t<float> a({f.height(), f.width()});
        std::copy(d->begin(), d->end(), a.mutable_data());
        return std::move(a);
      })
      .def("depth", [](DepthFrame& f) -> py::object {
        const std::vector<float>* d = f.sourceDepth();
        if (!d) return py::none();
        const std::vector<float> x = f.depthXform().apply(*d, f.width(), f.height());
        py::array_t<float> a({f.height(), f.width()});
        std::copy(x.begin(), x.end(), a.mutable_data());
        return std::move(a);
      })
      .def("setDepth", &DepthFrame::setDepth)
      .def("warp", [](DepthFrame& f) { return f.spatialXform().warp(f.height(), f.width()); })
      .def("clear", &DepthFrame::clear).def("clearCache", &DepthFrame::clearCache)
      .def("clearXformedCache", &DepthFrame::clearXformedCache)
      .def("depthXform", &DepthFrame::depthXform, py::return_value_policy::reference)
      .def("resetDepthXform", &DepthFrame::resetDepthXform)
      .def("spatialXform", &DepthFrame::spatialXform, py::return_value_policy::reference)
      .def("resetSpatialXform", &DepthFrame::resetSpatialXform)
      .def_readwrite("intrinsics", &DepthFrame::intrinsics)
      .def_readwrite("extrinsics", &DepthFrame::extrinsics);

  py::class_<DepthStream>(m, "DepthStream")
      .def("frame", &DepthStream::frame, py::return_value_policy::reference)
      .def("name", [](DepthStream& d) { return d.name_; }).def("path", [](DepthStream& d) { return d.path_; })
      .def("depthXformDesc", [](DepthStream& d) { return d.depthXformDesc_; })
      .def("spatialXformDesc", [](DepthStream& d) { return d.spatialXformDesc_; })
      .def("width", &DepthStream::width).def("height", &DepthStream::height)
      .def("setDir", &DepthStream::setDir)
      .def("resetDepthXforms", &DepthStream::resetDepthXforms)
      .def("resetSpatialXforms", &DepthStream::resetSpatialXforms)
      .def("clearCache", &DepthStream::clearCache);

  py::class_<DepthVideo>(m, "DepthVideo")
      .def(py::init())
      .def("printInfo", &DepthVideo::printInfo).def("reset", &DepthVideo::reset).def("save", &DepthVideo::save)
      .def("load", [](DepthVideo&, const std::string&) { throw std::runtime_error("DepthVideo::load: the reference's own load() cannot read the video.dat its save() writes (SURVEY.md appendix D); not provided."); })
      .def("width", [](DepthVideo& v) { return v.width_; }).def("height", [](DepthVideo& v) { return v.height_; })
      .def("aspect", [](DepthVideo& v) { return v.aspect_; }).def("invAspect", [](DepthVideo& v) { return v.invAspect_; })
      .def("path", [](DepthVideo& v) { return v.path_; })
      .def("numFrames", &DepthVideo::numFrames).def("duration", [](DepthVideo& v) { return v.duration_; })
      .def("numColorStreams", &DepthVideo::numColorStreams).def("hasColorStream", &DepthVideo::hasColorStream)
      .def("colorStreamIndex", &DepthVideo::colorStreamIndex)
      .def("colorStream", [](DepthVideo& v, int i) -> ColorStream& { return *v.colorStreams_.at(i); }, py::return_value_policy::reference)
      .def("colorStream", [](DepthVideo& v, const std::string& n) -> ColorStream& { return *v.colorStreams_.at(v.colorStreamIndex(n)); }, py::return_value_policy::reference)
      .def("createColorStream", &DepthVideo::createColorStream, py::arg("name"), py::arg("dir"), py::arg("extension"),
           py::arg("type"), py::arg("size") = std::pair<int, int>{-1, -1})
      .def("numDepthStreams", &DepthVideo::numDepthStreams).def("hasDepthStream", &DepthVideo::hasDepthStream)
      .def("depthStreamIndex", &DepthVideo::depthStreamIndex)
      .def("depthStream", [](DepthVideo& v, int i) -> DepthStream& { return *v.depthStreams_.at(i); }, py::return_value_policy::reference)
      .def("depthStream", [](DepthVideo& v, const std::string& n) -> DepthStream& { return *v.depthStreams_.at(v.depthStreamIndex(n)); }, py::return_value_policy::reference)
      .def("createDepthStream", &DepthVideo::createDepthStream, py::arg("name"), py::arg("dir"), py::arg("size") = std::pair<int, int>{-1, -1})
      .def("depthFrame", [](DepthVideo& v, int s, int f) -> DepthFrame& { return v.depthStreams_.at(s)->frame(f); }, py::return_value_policy::reference)
      .def("clearDepthCaches", &DepthVideo::clearDepthCaches)
      .def("saveDepth", [](DepthVideo&, int) { throw std::runtime_error("saveDepth is outside the optimizer path of this build."); });

  py::class_<FrameRange>(m, "FrameRange")
      .def(py::init())
      .def("fromString", &FrameRange::fromString).def("toString", &FrameRange::toString)
      .def("resolve", &FrameRange::resolve, py::arg("numFrames"), py::arg("clip") = false)
      .def("isEmpty", &FrameRange::isEmpty).def("firstFrame", &FrameRange::firstFrame).def("lastFrame", &FrameRange::lastFrame)
      .def("count", &FrameRange::count).def("isConsecutive", &FrameRange::isConsecutive).def("inRange", &FrameRange::inRange)
      .def("checkEmpty", &FrameRange::checkEmpty);

  py::class_<FlowConstraintsParams>(m, "FlowConstraintsParams")
      .def(py::init())
      .def_readwrite("matchSeparation", &FlowConstraintsParams::matchSeparation)
      .def_readwrite("minDynamicDistance", &FlowConstraintsParams::minDynamicDistance)
      .def_readwrite("frameRange", &FlowConstraintsParams::frameRange)
      .def_readwrite("doNotUseCache", &FlowConstraintsParams::doNotUseCache);
  py::class_<FlowConstraintsCollection>(m, "FlowConstraintsCollection")
      .def(py::init<const DepthVideo&, const FlowConstraintsParams&>(), py::keep_alive<1, 2>())
      .def("load", &FlowConstraintsCollection::load).def("save", &FlowConstraintsCollection::save)
      .def("resetStaticFlag", &FlowConstraintsCollection::resetStaticFlag)
      .def("setStaticFlagFromDynamicMask", &FlowConstraintsCollection::setStaticFlagFromDynamicMask)
      .def("pruneStaticFlag", &FlowConstraintsCollection::pruneStaticFlag)
      .def("setStaticFlags", &FlowConstraintsCollection::setStaticFlags)
      // extensions (test access): the isStatic flags and the (loc0.xy, loc1.xy) rows of one pair
      .def("staticFlags", [](const FlowConstraintsCollection& c, int a, int b) { return c.pairs_.at({a, b}).isStatic; })
      .def("pairLocations", [](const FlowConstraintsCollection& c, int a, int b) {
        const auto& loc = c.pairs_.at({a, b}).loc;
        py::array_t<float> out({static_cast<py::ssize_t>(loc.size()), static_cast<py::ssize_t>(4)});
        auto o = out.mutable_unchecked<2>();
        for (size_t i = 0; i < loc.size(); ++i)
          for (int k = 0; k < 4; ++k) o(i, k) = loc[i][k];
        return out;
      })
      .def("numPairs", &FlowConstraintsCollection::numPairs)
      .def("numConstraints", &FlowConstraintsCollection::numConstraints)
      .def("holdsFlowImages", [](const FlowConstraintsCollection& c) { return c.denseValid_; })  // (extension: dense hand-over possible)
      .def("compute", &FlowConstraintsCollection::compute, py::call_guard<py::gil_scoped_release>())
      .def_readwrite("device", &FlowConstraintsCollection::device_);

  py::class_<DepthVideoImporter>(m, "DepthVideoImporter")
      .def_static("importVideo", &DepthVideoImporter::importVideo)
      .def_static("importPoses", [](DepthVideo&, const std::string&, int) { throw std::runtime_error("importPoses is outside the optimizer path of this build."); })
      .def_static("importColmapRecon", [](py::args) { throw std::runtime_error("COLMAP import is outside the optimizer path of this build."); })
      .def_static("importColmapDepth", [](py::args) { throw std::runtime_error("COLMAP import is outside the optimizer path of this build."); });

  py::enum_<StaticLossType>(m, "StaticLossType").value("Euclidean", StaticLossType::Euclidean).value("ReproDisparity", StaticLossType::ReproDisparity)
      .value("ReproDepthRatio", StaticLossType::ReproDepthRatio).value("ReproLogDepth", StaticLossType::ReproLogDepth);
  py::enum_<SmoothLossType>(m, "SmoothLossType").value("EuclideanLaplacian", SmoothLossType::EuclideanLaplacian)
      .value("ReproDisparityLaplacian", SmoothLossType::ReproDisparityLaplacian)
      .value("ReproDepthRatioConsistency", SmoothLossType::ReproDepthRatioConsistency)
      .value("ReproLogDepthConsistency", SmoothLossType::ReproLogDepthConsistency);
  py::enum_<IntrinsicsOptimization>(m, "IntrinsicsOptimization").value("Fixed", IntrinsicsOptimization::Fixed)
      .value("Shared", IntrinsicsOptimization::Shared).value("PerFrame", IntrinsicsOptimization::PerFrame);

  struct DvpoTag {};
  py::class_<DvpoTag> dvpo(m, "DepthVideoPoseOptimizer");
  py::class_<DvpoParams>(dvpo, "Params")
      .def(py::init())
      .def(py::init<const DvpoParams&>())
      .def_readwrite("frameRange", &DvpoParams::frameRange).def_readwrite("maxIterations", &DvpoParams::maxIterations)
      .def_readwrite("numThreads", &DvpoParams::numThreads).def_readwrite("numSteps", &DvpoParams::numSteps)
      .def_readwrite("robustness", &DvpoParams::robustness).def_readwrite("huberLoss", &DvpoParams::huberLoss).def_readwrite("staticLossType", &DvpoParams::staticLossType)
      .def_readwrite("staticSpatialWeight", &DvpoParams::staticSpatialWeight).def_readwrite("staticDepthWeight", &DvpoParams::staticDepthWeight)
      .def_readwrite("smoothLossType", &DvpoParams::smoothLossType).def_readwrite("smoothStaticWeight", &DvpoParams::smoothStaticWeight)
      .def_readwrite("smoothDynamicWeight", &DvpoParams::smoothDynamicWeight).def_readwrite("positionReg", &DvpoParams::positionReg)
      .def_readwrite("scaleReg", &DvpoParams::scaleReg).def_readwrite("scaleRegGridSize", &DvpoParams::scaleRegGridSize)
      .def_readwrite("depthDeformRegInitial", &DvpoParams::depthDeformRegInitial).def_readwrite("depthDeformRegFinal", &DvpoParams::depthDeformRegFinal)
      .def_readwrite("adaptiveDeformationCost", &DvpoParams::adaptiveDeformationCost).def_readwrite("spatialDeformReg", &DvpoParams::spatialDeformReg)
      .def_readwrite("graduateDepthDeformReg", &DvpoParams::graduateDepthDeformReg).def_readwrite("focalReg", &DvpoParams::focalReg)
      .def_readwrite("coarseToFine", &DvpoParams::coarseToFine).def_readwrite("ctfLong", &DvpoParams::ctfLong)
      .def_readwrite("ctfShort", &DvpoParams::ctfShort).def_readwrite("deferredSpatialOpt", &DvpoParams::deferredSpatialOpt)
      .def_readwrite("dsoLong", &DvpoParams::dsoLong).def_readwrite("dsoShort", &DvpoParams::dsoShort)
      .def_readwrite("focalLong", &DvpoParams::focalLong).def_readwrite("intrOpt", &DvpoParams::intrOpt)
      .def_readwrite("fixPoses", &DvpoParams::fixPoses).def_readwrite("fixDepthXforms", &DvpoParams::fixDepthXforms)
      .def_readwrite("fixSpatialXforms", &DvpoParams::fixSpatialXforms);

  py::class_<DepthVideoProcessor> dvp(m, "DepthVideoProcessor");
  py::class_<DvpParams>(dvp, "Params")
      .def(py::init())
      .def_readwrite("op", &DvpParams::op).def_readwrite("frameRange", &DvpParams::frameRange)
      .def_readwrite("colorStream", &DvpParams::colorStream).def_readwrite("depthStream", &DvpParams::depthStream)
      .def_readwrite("sourceDepthStream", &DvpParams::sourceDepthStream).def_readwrite("spatialRadius", &DvpParams::spatialRadius)
      .def_readwrite("frameRadius", &DvpParams::frameRadius).def_readwrite("depthSigma", &DvpParams::depthSigma)
      .def_readwrite("colorSigma", &DvpParams::colorSigma).def_readwrite("median", &DvpParams::median)
      .def_readwrite("farConnections", &DvpParams::farConnections).def_readwrite("maxDepth", &DvpParams::maxDepth).def_readwrite("matchSeparation", &DvpParams::matchSeparation)
      .def_readwrite("flowConsistancyThresh", &DvpParams::flowConsistancyThresh)
      .def_readwrite("trackSpawnDistance", &DvpParams::trackSpawnDistance).def_readwrite("trackPruneDistance", &DvpParams::trackPruneDistance)
      .def_readwrite("minDynamicDistance", &DvpParams::minDynamicDistance).def_readwrite("minTrackLength", &DvpParams::minTrackLength)
      .def_readwrite("depthXformDesc", &DvpParams::depthXformDesc).def_readwrite("spatialXformDesc", &DvpParams::spatialXformDesc)
      .def_readwrite("poseOptimizer", &DvpParams::poseOptimizer);
  py::enum_<Op>(dvp, "Op")
      .value("None", Op::None).value("Reset", Op::Reset).value("Copy", Op::Copy).value("BilateralFilter", Op::BilateralFilter)
      .value("FlowGuidedFilter", Op::FlowGuidedFilter).value("ComputeConstraints", Op::ComputeConstraints)
      .value("ResetConstraintStaticFlag", Op::ResetConstraintStaticFlag)
      .value("SetConstraintStaticFlagFromDynamicMask", Op::SetConstraintStaticFlagFromDynamicMask)
      .value("ComputeTracks", Op::ComputeTracks).value("GridXformSplit", Op::GridXformSplit).value("ResetPoses", Op::ResetPoses)
      .value("ResetDepthXforms", Op::ResetDepthXforms).value("ResetSpatialXforms", Op::ResetSpatialXforms)
      .value("NormalizeDepth", Op::NormalizeDepth).value("OptimizePoses", Op::OptimizePoses)
      .value("ResetNormalizeOptimize", Op::ResetNormalizeOptimize)
      .value("ClipMaxDepth", Op::ClipMaxDepth)                              // (extensions: C++-only in the reference)
      .value("PruneConstraintStaticFlag", Op::PruneConstraintStaticFlag);
  dvp.def(py::init<DepthVideo*>(), py::keep_alive<1, 2>())
      .def("process", &DepthVideoProcessor::process).def("reset", &DepthVideoProcessor::reset)
      .def("gridXformSplit", &DepthVideoProcessor::gridXformSplit).def("resetPoses", &DepthVideoProcessor::resetPoses)
      .def("resetDepthXforms", &DepthVideoProcessor::resetDepthXforms).def("resetSpatialXforms", &DepthVideoProcessor::resetSpatialXforms)
      .def("normalizeDepth", &DepthVideoProcessor::normalizeDepth).def("optimizePoses", &DepthVideoProcessor::optimizePoses)
      .def_readwrite("device", &DepthVideoProcessor::device_)
      .def_readonly("usedFlowImages", &DepthVideoProcessor::usedFlowImages_);  // (extension, with `device`)
}
