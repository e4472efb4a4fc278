// host_registration.hpp — the host-side neighbours of the hot path behind
// the C ABI: PLY vertex reader, synthetic putative associations, precision / recall, closed-form
// rigid transform. No device code. Part of clipper_hip.hip (one translation unit).
//   reference: benchmarks/bm_utils.cpp:24-79 (read_ply), :277-341 (generate_synthetic_
//   correspondences), :345-371 (get_precision_recall); the transform estimate closes the loop of
//   examples/python/ex4_bunny.ipynb cells 6-8 (Arun / Horn: centroids, 3x3 cross-covariance, SVD).
#pragma once

#include <fstream>
#include <random>
#include <set>
#include <sstream>
#include <unordered_set>

namespace {

struct PlyProp {
  std::string name;
  int bytes;       // size of the scalar
  int kind;        // 0 = signed int, 1 = unsigned int, 2 = float
};

int ply_type(const std::string& t, PlyProp& p) {
  static const struct { const char* n; int b, k; } T[] = {
      {"char", 1, 0},   {"int8", 1, 0},   {"uchar", 1, 1},  {"uint8", 1, 1},  {"short", 2, 0},
      {"int16", 2, 0},  {"ushort", 2, 1}, {"uint16", 2, 1}, {"int", 4, 0},    {"int32", 4, 0},
      {"uint", 4, 1},   {"uint32", 4, 1}, {"float", 4, 2},  {"float32", 4, 2}, {"double", 8, 2},
      {"float64", 8, 2}};
  for (const auto& e : T)
    if (t == e.n) {
      p.bytes = e.b;
      p.kind = e.k;
      return 0;
    }
  return -1;
}

double ply_scalar(const unsigned char* b, const PlyProp& p, bool swap) {
  unsigned char t[8];
  for (int i = 0; i < p.bytes; ++i) t[i] = swap ? b[p.bytes - 1 - i] : b[i];
  if (p.kind == 2) {
    if (p.bytes == 4) { float f; std::memcpy(&f, t, 4); return f; }
    double d; std::memcpy(&d, t, 8); return d;
  }
  if (p.bytes == 1) return p.kind ? static_cast<double>(t[0]) : static_cast<double>(static_cast<signed char>(t[0]));
  if (p.bytes == 2) { uint16_t v; std::memcpy(&v, t, 2); return p.kind ? static_cast<double>(v) : static_cast<double>(static_cast<int16_t>(v)); }
  uint32_t v; std::memcpy(&v, t, 4);
  return p.kind ? static_cast<double>(v) : static_cast<double>(static_cast<int32_t>(v));
}

// x, y, z of the `vertex` element (ascii or binary, either endianness; other scalar properties
// are skipped, a list property inside the vertex element is an error) -> pts (n x 3, row-major)
int read_ply_vertices(const char* path, std::vector<double>& pts, int64_t& n) {
  std::ifstream f(path, std::ios::binary);
  if (!f) return fail(CLIPPER_HIP_E_INVALID, "cannot open %s", path);
  std::string line;
  std::getline(f, line);
  while (!line.empty() && (line.back() == '\r' || line.back() == ' ')) line.pop_back();
  if (line != "ply") return fail(CLIPPER_HIP_E_INVALID, "%s is not a PLY file", path);
  std::string fmt;
  std::vector<PlyProp> props;
  bool in_vertex = false, seen_vertex = false, ended = false;
  n = 0;
  while (std::getline(f, line)) {
    std::istringstream ss(line);
    std::string tok;
    if (!(ss >> tok)) continue;
    if (tok == "format") ss >> fmt;
    else if (tok == "element") {
      std::string name;
      long long cnt = 0;
      ss >> name >> cnt;
      in_vertex = (name == "vertex");
      if (in_vertex) {
        if (seen_vertex) return fail(CLIPPER_HIP_E_INVALID, "two vertex elements");
        seen_vertex = true;
        n = cnt;
      } else if (!seen_vertex) {
        return fail(CLIPPER_HIP_E_INVALID, "elements before `vertex` are not supported");
      }
    } else if (tok == "property" && in_vertex) {
      std::string type, name;
      ss >> type;
      if (type == "list") return fail(CLIPPER_HIP_E_INVALID, "list property in the vertex element");
      ss >> name;
      PlyProp p;
      p.name = name;
      if (ply_type(type, p)) return fail(CLIPPER_HIP_E_INVALID, "unknown PLY type %s", type.c_str());
      props.push_back(p);
    } else if (tok == "end_header") {
      ended = true;
      break;
    }
  }
  if (!ended) return fail(CLIPPER_HIP_E_INVALID, "PLY header not terminated");
  int ix[3] = {-1, -1, -1};
  for (size_t i = 0; i < props.size(); ++i)
    for (int k = 0; k < 3; ++k)
      if (props[i].name == std::string(1, "xyz"[k])) ix[k] = static_cast<int>(i);
  if (ix[0] < 0 || ix[1] < 0 || ix[2] < 0) return fail(CLIPPER_HIP_E_INVALID, "vertex element has no x, y, z");
  if (n < 0) return fail(CLIPPER_HIP_E_INVALID, "negative vertex count");
  {
    // the header's count is not trusted: every vertex takes at least one byte per property of the body
    // (ascii: a digit and a separator each), so a count the rest of the file cannot hold is refused
    // before anything is allocated for it
    const std::streampos body = f.tellg();
    f.seekg(0, std::ios::end);
    const std::streamoff left = f.tellg() - body;
    f.seekg(body);
    size_t per = 0;
    for (const PlyProp& pp : props) per += (fmt == "ascii") ? 2u : static_cast<size_t>(pp.bytes);
    if (left < 0 || static_cast<unsigned long long>(n) * std::max<size_t>(per, 1) > static_cast<unsigned long long>(left) + 1ull)
      return fail(CLIPPER_HIP_E_INVALID, "PLY header declares %lld vertices, the file holds %lld more bytes",
                  static_cast<long long>(n), static_cast<long long>(left));
  }
  pts.assign(static_cast<size_t>(n) * 3, 0.0);
  if (fmt == "ascii") {
    std::vector<double> row(props.size());
    for (int64_t i = 0; i < n; ++i) {
      if (!std::getline(f, line)) return fail(CLIPPER_HIP_E_INVALID, "PLY body truncated at vertex %lld", static_cast<long long>(i));
      std::istringstream ss(line);
      for (size_t k = 0; k < props.size(); ++k)
        if (!(ss >> row[k])) return fail(CLIPPER_HIP_E_INVALID, "PLY body malformed at vertex %lld", static_cast<long long>(i));
      for (int k = 0; k < 3; ++k) {  // through the declared type, as a binary file would carry it
        const PlyProp& pp = props[static_cast<size_t>(ix[k])];
        double v = row[static_cast<size_t>(ix[k])];
        if (pp.kind == 2 && pp.bytes == 4) v = static_cast<double>(static_cast<float>(v));
        else if (pp.kind != 2) v = std::trunc(v);
        pts[static_cast<size_t>(i) * 3 + k] = v;
      }
    }
  } else if (fmt == "binary_little_endian" || fmt == "binary_big_endian") {
    const uint16_t probe = 1;
    const bool host_little = *reinterpret_cast<const unsigned char*>(&probe) == 1;
    const bool swap = (fmt == "binary_little_endian") != host_little;
    size_t stride = 0;
    std::vector<size_t> off(props.size());
    for (size_t k = 0; k < props.size(); ++k) {
      off[k] = stride;
      stride += static_cast<size_t>(props[k].bytes);
    }
    std::vector<unsigned char> buf(stride * static_cast<size_t>(n));
    f.read(reinterpret_cast<char*>(buf.data()), static_cast<std::streamsize>(buf.size()));
    if (static_cast<size_t>(f.gcount()) != buf.size()) return fail(CLIPPER_HIP_E_INVALID, "PLY body truncated");
    for (int64_t i = 0; i < n; ++i)
      for (int k = 0; k < 3; ++k)
        pts[static_cast<size_t>(i) * 3 + k] =
            ply_scalar(buf.data() + static_cast<size_t>(i) * stride + off[static_cast<size_t>(ix[k])],
                       props[static_cast<size_t>(ix[k])], swap);
  } else {
    return fail(CLIPPER_HIP_E_INVALID, "unknown PLY format %s", fmt.c_str());
  }
  return 0;
}

// 3x3 SVD by one-sided Jacobi on H = U S V^T (columns of U, V); returns det(U V^T)'s sign through R
// Horn / Arun: R = U diag(1, 1, det(U V^T)) V^T maximises trace(R^T H).
void rotation_from_cross_covariance(const double H[9], double R[9]) {
  // one-sided Jacobi: rotate the columns of A = H until they are orthogonal; V accumulates
  double A[9], V[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  std::memcpy(A, H, sizeof(A));
  for (int sweep = 0; sweep < 60; ++sweep) {
    double offn = 0.0;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        double a = 0, b = 0, c = 0;
        for (int i = 0; i < 3; ++i) {
          a += A[i * 3 + p] * A[i * 3 + p];
          b += A[i * 3 + q] * A[i * 3 + q];
          c += A[i * 3 + p] * A[i * 3 + q];
        }
        offn = std::max(offn, std::fabs(c) / (std::sqrt(a * b) + 1e-300));
        if (std::fabs(c) <= 1e-300) continue;
        const double zeta = (b - a) / (2.0 * c);
        const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
        const double cs = 1.0 / std::sqrt(1.0 + t * t), sn = cs * t;
        for (int i = 0; i < 3; ++i) {
          const double ap = A[i * 3 + p], aq = A[i * 3 + q];
          A[i * 3 + p] = cs * ap - sn * aq;
          A[i * 3 + q] = sn * ap + cs * aq;
          const double vp = V[i * 3 + p], vq = V[i * 3 + q];
          V[i * 3 + p] = cs * vp - sn * vq;
          V[i * 3 + q] = sn * vp + cs * vq;
        }
      }
    if (offn < 1e-15) break;
  }
  // singular values = column norms of A, U = A / s; order does not matter for U V^T, but the
  // reflection fix must act on the SMALLEST singular direction
  double s[3], U[9];
  int smallest = 0;
  for (int j = 0; j < 3; ++j) {
    s[j] = std::sqrt(A[j] * A[j] + A[3 + j] * A[3 + j] + A[6 + j] * A[6 + j]);
    if (s[j] < s[smallest]) smallest = j;
  }
  for (int j = 0; j < 3; ++j)
    for (int i = 0; i < 3; ++i) U[i * 3 + j] = s[j] > 1e-300 ? A[i * 3 + j] / s[j] : 0.0;
  // a vanishing singular value leaves its column of U undetermined: complete it by a cross product
  if (s[smallest] <= 1e-300 * (s[0] + s[1] + s[2]) || s[smallest] == 0.0) {
    const int a = (smallest + 1) % 3, b = (smallest + 2) % 3;
    U[0 * 3 + smallest] = U[1 * 3 + a] * U[2 * 3 + b] - U[2 * 3 + a] * U[1 * 3 + b];
    U[1 * 3 + smallest] = U[2 * 3 + a] * U[0 * 3 + b] - U[0 * 3 + a] * U[2 * 3 + b];
    U[2 * 3 + smallest] = U[0 * 3 + a] * U[1 * 3 + b] - U[1 * 3 + a] * U[0 * 3 + b];
  }
  auto det3 = [](const double* M) {
    return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) +
           M[2] * (M[3] * M[7] - M[4] * M[6]);
  };
  const double sgn = det3(U) * det3(V) < 0 ? -1.0 : 1.0;  // det(U V^T)
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double r = 0.0;
      for (int k = 0; k < 3; ++k) r += U[i * 3 + k] * (k == smallest ? sgn : 1.0) * V[j * 3 + k];
      R[i * 3 + j] = r;
    }
}

}  // namespace

extern "C" {

int64_t clipper_hip_read_ply_xyz(const char* path, double* pts_out, int64_t capacity) try {
  if (!path) return fail(CLIPPER_HIP_E_INVALID, "invalid argument");
  std::vector<double> pts;
  int64_t n = 0;
  try {  // (no exception may cross the C boundary: a file too large for this host's memory is an error code)
    if (int rc = read_ply_vertices(path, pts, n)) return rc;
  } catch (const std::exception& e) {
    return fail(CLIPPER_HIP_E_NOMEM, "reading %s: %s", path, e.what());
  }
  if (pts_out == nullptr) return n;  // size query
  if (capacity < n) return fail(CLIPPER_HIP_E_INVALID, "capacity %lld < %lld vertices", static_cast<long long>(capacity), static_cast<long long>(n));
  // column-major 3 x n (clipper::invariants::Data): datum i = pts_out[3 i .. 3 i + 2]
  std::memcpy(pts_out, pts.data(), pts.size() * sizeof(double));
  return n;
} CLIPPER_HIP_GUARD_INT

int clipper_hip_generate_synthetic_correspondences(int64_t n0, int64_t n1, const int32_t* Agood,
                                                   int64_t p, int64_t m, double rho, uint64_t seed,
                                                   int32_t* A_out, int32_t* Agt_out,
                                                   int64_t* ni_out) try {
  if (n0 < 1 || n1 < 1 || m < 1 || p < 0 || (p > 0 && !Agood) || !A_out || !Agt_out)
    return fail(CLIPPER_HIP_E_INVALID, "invalid argument");
  if (!(rho >= 0.0 && rho <= 1.0)) return fail(CLIPPER_HIP_E_INVALID, "outlier ratio must be in [0, 1]");
  const int64_t ni = static_cast<int64_t>(std::llround(static_cast<double>(m) * (1.0 - rho)));  // :286
  const int64_t no = m - ni;                                                                    // :289
  if (ni > p)  // :294-298
    return fail(CLIPPER_HIP_E_STATE, "not enough initial inliers (%lld) for the requested outlier ratio (%g, %lld)",
                static_cast<long long>(p), rho, static_cast<long long>(ni));
  if (static_cast<double>(no) > static_cast<double>(n0) * static_cast<double>(n1) - static_cast<double>(p))
    return fail(CLIPPER_HIP_E_INVALID, "more outliers requested than there are wrong pairs");
  for (int64_t i = 0; i < p; ++i)
    if (Agood[i] < 0 || Agood[i] >= n0 || Agood[p + i] < 0 || Agood[p + i] >= n1)
      return fail(CLIPPER_HIP_E_INVALID, "Agood row %lld out of range", static_cast<long long>(i));
  std::mt19937_64 rng(seed);  // the reference seeds from random_device (:300-301): not reproducible
  // inliers: ni of the p good associations without replacement (:307-315), placed LAST
  std::vector<int64_t> I(static_cast<size_t>(p));
  for (int64_t i = 0; i < p; ++i) I[static_cast<size_t>(i)] = i;
  for (int64_t i = p - 1; i > 0; --i) {  // Fisher-Yates with our own draw (std::shuffle is not portable)
    std::uniform_int_distribution<int64_t> d(0, i);
    std::swap(I[static_cast<size_t>(i)], I[static_cast<size_t>(d(rng))]);
  }
  for (int64_t i = 0; i < ni; ++i) {  // column-major m x 2 / ni x 2
    const int64_t g = I[static_cast<size_t>(i)];
    Agt_out[i] = Agood[g];
    Agt_out[ni + i] = Agood[p + g];
    A_out[no + i] = Agood[g];
    A_out[m + no + i] = Agood[p + g];
  }
  // outliers: uniformly from all n0 * n1 pairs, never twice, never a good one (:318-338), FIRST
  std::unordered_set<int64_t> good, tried;
  good.reserve(static_cast<size_t>(p) * 2);
  for (int64_t i = 0; i < p; ++i) good.insert(static_cast<int64_t>(Agood[i]) * n1 + Agood[p + i]);
  std::uniform_int_distribution<int64_t> dis(0, n0 * n1 - 1);
  int64_t nele = 0;
  while (nele < no) {
    const int64_t k = dis(rng);
    if (!tried.insert(k).second) continue;
    if (good.count(k)) continue;
    A_out[nele] = static_cast<int32_t>(k / n1);      // k2ij_full (:268-273)
    A_out[m + nele] = static_cast<int32_t>(k % n1);
    ++nele;
  }
  if (ni_out) *ni_out = ni;
  return 0;
} CLIPPER_HIP_GUARD_INT

int clipper_hip_precision_recall(const int32_t* A, int64_t na, const int32_t* Agt, int64_t ngt,
                                 double* precision, double* recall) try {
  if (!precision || !recall || na < 0 || ngt < 0 || (na > 0 && !A) || (ngt > 0 && !Agt))
    return fail(CLIPPER_HIP_E_INVALID, "invalid argument");
  *precision = *recall = 0.0;
  if (na == 0 || ngt == 0) return 0;  // :352
  std::set<std::pair<int32_t, int32_t>> gt;
  for (int64_t i = 0; i < ngt; ++i) gt.insert({Agt[i], Agt[ngt + i]});
  int64_t TP = 0;  // rows of A counted one by one, repeated rows included (:354-357)
  for (int64_t i = 0; i < na; ++i) TP += gt.count({A[i], A[na + i]}) ? 1 : 0;
  *precision = static_cast<double>(TP) / static_cast<double>(na);
  *recall = static_cast<double>(TP) / static_cast<double>(ngt);
  return 0;
} CLIPPER_HIP_GUARD_INT

int clipper_hip_estimate_rigid_transform(const double* D1, int64_t n1, const double* D2, int64_t n2,
                                         const int32_t* A, int64_t k, double* T_out) try {
  if (!D1 || !D2 || !A || !T_out || k < 3)
    return fail(CLIPPER_HIP_E_INVALID, "at least 3 associations and non-null arguments are needed");
  double cp[3] = {0, 0, 0}, cq[3] = {0, 0, 0};
  for (int64_t i = 0; i < k; ++i) {
    const int32_t a = A[i], b = A[k + i];
    if (a < 0 || a >= n1 || b < 0 || b >= n2) return fail(CLIPPER_HIP_E_INVALID, "association %lld out of range", static_cast<long long>(i));
    for (int c = 0; c < 3; ++c) {
      cp[c] += D1[3 * a + c];
      cq[c] += D2[3 * b + c];
    }
  }
  for (int c = 0; c < 3; ++c) {
    cp[c] /= static_cast<double>(k);
    cq[c] /= static_cast<double>(k);
  }
  double H[9] = {0};  // H = sum (q - cq)(p - cp)^T
  for (int64_t i = 0; i < k; ++i) {
    const int32_t a = A[i], b = A[k + i];
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) H[r * 3 + c] += (D2[3 * b + r] - cq[r]) * (D1[3 * a + c] - cp[c]);
  }
  double R[9];
  rotation_from_cross_covariance(H, R);
  // column-major 4 x 4
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c) T_out[c * 4 + r] = (r == c) ? 1.0 : 0.0;
  for (int r = 0; r < 3; ++r) {
    double t = cq[r];
    for (int c = 0; c < 3; ++c) {
      T_out[c * 4 + r] = R[r * 3 + c];
      t -= R[r * 3 + c] * cp[c];
    }
    T_out[3 * 4 + r] = t;
  }
  return 0;
} CLIPPER_HIP_GUARD_INT

}  // extern "C"
