// test_facade.cpp — the reference's own hot-path tests (test/affinity_test.cpp:14-108,
// test/clipper_test.cpp:15-68 and the get/set round trip of :72-133), restated against the
// clipper::CLIPPER facade of this build. Plain asserts instead of gtest (not in the image).
// Built and run on the GPU box by tests/test_gpu_clipperpy.py.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <memory>

#include <clipper/clipper.h>
#include <clipper/dsd.h>
#include <clipper/utils.h>

#define EXPECT(cond)                                                        \
  do {                                                                      \
    if (!(cond)) {                                                          \
      std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond);         \
      std::exit(1);                                                         \
    }                                                                       \
  } while (0)

namespace {

// affinity_test.cpp:33-48 — 4 model points, rotated pi/8 about z, translated (5,3,0), the
// last data point dropped
void make_data(clipper::invariants::Data& model, clipper::invariants::Data& data) {
  model = clipper::invariants::Data::Zero(3, 4);
  const double pts[4][3] = {{0, 0, 0}, {2, 0, 0}, {0, 3, 0}, {2, 2, 0}};
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 3; ++r) model(r, c) = pts[c][r];
  const double th = M_PI / 8, t[3] = {5, 3, 0};
  const double R[3][3] = {{std::cos(th), -std::sin(th), 0}, {std::sin(th), std::cos(th), 0}, {0, 0, 1}};
  data = clipper::invariants::Data::Zero(3, 3);  // T_MD.inverse() * model, first 3 columns
  for (int c = 0; c < 3; ++c)
    for (int r = 0; r < 3; ++r) {
      double acc = 0;
      for (int k = 0; k < 3; ++k) acc += R[k][r] * (model(k, c) - t[k]);
      data(r, c) = acc;
    }
}

const double Mtrue[12][12] = {  // affinity_test.cpp:93-106
    {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0}, {0, 1, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0},
    {0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 0, 0}, {0, 1, 0, 1, 0, 0, 0, 0, 0, 0, 1, 0},
    {1, 0, 0, 0, 1, 0, 0, 0, 1, 1, 0, 0}, {0, 0, 0, 0, 0, 1, 0, 1, 0, 0, 0, 0},
    {0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 1, 0, 1, 0, 0, 0, 0},
    {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0}, {0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0},
    {0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 1, 0}, {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1}};

}  // namespace

int main() {
  clipper::invariants::EuclideanDistance::Params iparams;
  clipper::invariants::EuclideanDistancePtr invariant =
      std::make_shared<clipper::invariants::EuclideanDistance>(iparams);
  clipper::Params params;
  clipper::invariants::Data model, data;
  make_data(model, data);

  // ---- TEST(Affinity, EuclideanDistance) ------------------------------------------------
  {
    clipper::CLIPPER clipper(invariant, params);
    clipper.scorePairwiseConsistency(model, data);  // all-to-all
    clipper::Association A = clipper.getInitialAssociations();
    const int n = static_cast<int>(model.cols() * data.cols());
    EXPECT(A.rows() == n);
    EXPECT(A.cols() == 2);
    for (int i = 0; i < model.cols(); i++)
      for (int j = 0; j < data.cols(); j++) {
        const int k = i * static_cast<int>(data.cols()) + j;
        EXPECT(A(k, 0) == i);
        EXPECT(A(k, 1) == j);
      }
    clipper::Affinity M = clipper.getAffinityMatrix();
    clipper::Constraint C = clipper.getConstraintMatrix();
    EXPECT(M.rows() == A.rows() && M.cols() == A.rows());
    for (int i = 0; i < n; ++i) EXPECT(M(i, i) == 1.0);
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j) {
        EXPECT(M(i, j) == M(j, i));
        EXPECT(C(i, j) == C(j, i));
        EXPECT(M(i, j) == C(i, j));
        EXPECT(M(i, j) == Mtrue[i][j]);
      }
    std::printf("Affinity.EuclideanDistance ok\n");
  }

  // ---- TEST(CLIPPER, EuclideanDistance) — random u0 as in the reference, plus a fixed one --
  {
    clipper::CLIPPER clipper(invariant, params);
    clipper.scorePairwiseConsistency(model, data);
    clipper::VectorXd u0(12);
    for (int i = 0; i < 12; ++i) u0(i) = 1.0 / std::sqrt(12.0);
    clipper.solve(u0);
    clipper::Association Ainliers = clipper.getSelectedAssociations();
    EXPECT(Ainliers.rows() == 3);
    for (int i = 0; i < Ainliers.rows(); ++i) EXPECT(Ainliers(i, 0) == Ainliers(i, 1));
    EXPECT(std::fabs(clipper.getSolution().score - 3.0) < 1e-6);
    int ok = 0;
    for (int trial = 0; trial < 20; ++trial) {  // clipper_test.cpp:54-66 with its random u0
      clipper.solve();
      clipper::Association Ain = clipper.getSelectedAssociations();
      bool good = Ain.rows() == 3;
      for (int i = 0; good && i < Ain.rows(); ++i) good = Ain(i, 0) == Ain(i, 1);
      ok += good;
    }
    EXPECT(ok >= 14);
    std::printf("CLIPPER.EuclideanDistance ok (%d/20 random u0 reach the 3-clique)\n", ok);
  }

  // ---- multi-GPU through the class: column shards over a device list, here two logical shards on
  // device 0 — the same answer as one shard, no launcher ----------------------------------------------------
  {
    clipper::CLIPPER sharded(invariant, params);
    sharded.setDevices({0, 0});
    sharded.scorePairwiseConsistency(model, data);
    clipper::CLIPPER single(invariant, params);
    single.scorePairwiseConsistency(model, data);
    clipper::VectorXd u0(12);
    for (int i = 0; i < 12; ++i) u0(i) = 1.0 / std::sqrt(12.0);
    sharded.solve(u0);
    single.solve(u0);
    EXPECT(sharded.getSolution().nodes == single.getSolution().nodes);
    EXPECT(std::fabs(sharded.getSolution().score - single.getSolution().score) < 1e-9);
    clipper::Affinity Ms = sharded.getAffinityMatrix(), M1 = single.getAffinityMatrix();
    for (int i = 0; i < 12; ++i)
      for (int j = 0; j < 12; ++j) EXPECT(Ms(i, j) == M1(i, j));
    bool threw = false;
    try {
      sharded.setDevices({0});  // (after the first GPU call: refused, like setDevice)
    } catch (const std::logic_error&) {
      threw = true;
    }
    EXPECT(threw);
    std::printf("CLIPPER.setDevices ok (two logical shards on one device)\n");
  }

  // ---- get/set round trip (clipper_test.cpp:115-133), solved with solve() ------------------
  {
    clipper::CLIPPER clipper(invariant, params);
    clipper.scorePairwiseConsistency(model, data);
    clipper::Affinity M = clipper.getAffinityMatrix();
    clipper::Constraint C = clipper.getConstraintMatrix();
    clipper::CLIPPER clipper2(invariant, params);
    clipper2.setMatrixData(M, C);
    clipper::VectorXd u0(12);
    for (int i = 0; i < 12; ++i) u0(i) = 1.0 / std::sqrt(12.0);
    clipper2.solve(u0);
    clipper::Association Ainliers = clipper::utils::selectInlierAssociations(
        clipper2.getSolution(), clipper.getInitialAssociations());
    EXPECT(Ainliers.rows() == 3);
    for (int i = 0; i < Ainliers.rows(); ++i) EXPECT(Ainliers(i, 0) == Ainliers(i, 1));
    // the SDP path the reference uses here is not built: it must say so and return nothing
    clipper2.solveAsMSRCSDR();
    EXPECT(clipper2.getSolution().nodes.empty() && clipper2.getSolution().score == -1);
    std::printf("CLIPPER.EuclideanDistance_UseGetSet ok\n");
  }

  // ---- a user-defined C++ invariant takes the host-evaluated route --------------------------
  {
    struct Binary : clipper::invariants::PairwiseInvariant {
      double operator()(const clipper::invariants::Datum& ai, const clipper::invariants::Datum& aj,
                        const clipper::invariants::Datum& bi,
                        const clipper::invariants::Datum& bj) override {
        double l1 = 0, l2 = 0;
        for (int k = 0; k < ai.size(); ++k) {
          l1 += (ai(k) - aj(k)) * (ai(k) - aj(k));
          l2 += (bi(k) - bj(k)) * (bi(k) - bj(k));
        }
        return std::fabs(std::sqrt(l1) - std::sqrt(l2)) < 0.06 ? 1.0 : 0.0;
      }
    };
    clipper::CLIPPER clipper(std::make_shared<Binary>(), params);
    clipper.scorePairwiseConsistency(model, data);
    clipper::Affinity M = clipper.getAffinityMatrix();
    for (int i = 0; i < 12; ++i)
      for (int j = 0; j < 12; ++j) EXPECT(M(i, j) == Mtrue[i][j]);
    std::printf("custom C++ invariant ok\n");
  }
  // ---- TEST(DSD, Solve) / TEST(DSD, SolveRestrictedGraph) — test/dsd_test.cpp:14-80, and the
  //      same matrix through CLIPPER::solve with Rounding::DSD ----------------------------------
  {
    static const double Mdsd[20][20] = {
      {1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.2964, 0.0},
      {0.0, 1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0138, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0},
      {0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0016, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0747, 0.0},
      {0.0, 0.0, 0.0, 1.0, 0.0, 0.0555, 0.2547, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0102, 0.0, 0.7715, 0.0, 0.0, 0.0, 0.0},
      {0.0, 0.0, 0.0, 0.0, 1.0, 0.0063, 0.0, 0.3846, 0.0, 0.0003, 0.0014, 0.0, 0.0, 0.0, 0.0, 0.0063, 0.0, 0.0, 0.0, 0.0},
      {0.0, 0.0, 0.0, 0.0555, 0.0063, 1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.9927, 0.0, 0.0, 0.9722, 0.0, 0.0, 0.0, 0.0},
      {0.0, 0.0, 0.0, 0.2547, 0.0, 0.0, 1.0, 0.0, 0.0023, 0.0, 0.0, 0.8775, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0},
      {0.0, 0.0, 0.0, 0.0, 0.3846, 0.0, 0.0, 1.0, 0.0001, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0},
      {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0023, 0.0001, 1.0, 0.7914, 0.0, 0.0, 0.0, 0.0617, 0.0, 0.0, 0.9938, 0.0, 0.0, 0.0007},
      {0.0, 0.0, 0.0, 0.0, 0.0003, 0.0, 0.0, 0.0, 0.7914, 1.0, 0.0, 0.0, 0.0001, 0.0091, 0.0, 0.2503, 0.0222, 0.0549, 0.0, 0.0},
      {0.0, 0.0, 0.0, 0.0, 0.0014, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0008},
      {0.0, 0.0, 0.0016, 0.0, 0.0, 0.0, 0.8775, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.7007, 0.0},
      {0.0, 0.0, 0.0, 0.0, 0.0, 0.9927, 0.0, 0.0, 0.0, 0.0001, 0.0, 0.0, 1.0, 0.0, 0.9978, 0.0, 0.0, 0.0, 0.0, 0.0},
      {0.0, 0.0138, 0.0, 0.0102, 0.0, 0.0, 0.0, 0.0, 0.0617, 0.0091, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0003, 0.0, 0.0},
      {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.9978, 0.0, 1.0, 0.0012, 0.0, 0.0, 0.0, 0.0074},
      {0.0, 0.0, 0.0, 0.7715, 0.0063, 0.9722, 0.0, 0.0, 0.0, 0.2503, 0.0, 0.0, 0.0, 0.0, 0.0012, 1.0, 0.0026, 0.0217, 0.0, 0.0},
      {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.9938, 0.0222, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0026, 1.0, 0.0, 0.0, 0.0},
      {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0549, 0.0, 0.0, 0.0, 0.0003, 0.0, 0.0217, 0.0, 1.0, 0.0007, 0.0},
      {0.2964, 0.0, 0.0747, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.7007, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0007, 1.0, 0.0},
      {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0007, 0.0, 0.0008, 0.0, 0.0, 0.0, 0.0074, 0.0, 0.0, 0.0, 0.0, 1.0}};
    clipper::MatrixXd M(20, 20);
    for (int i = 0; i < 20; ++i)
      for (int j = 0; j < 20; ++j) M(i, j) = Mdsd[i][j];
    const std::vector<int> truth = {3, 5, 12, 14, 15};
    EXPECT(clipper::dsd::solve(M) == truth);
    EXPECT(clipper::dsd::solve(M, {0, 1, 3, 5, 7, 12, 14, 15, 19}) == truth);
    // the exact rounding inside the solver: u's support contains the dense cluster
    clipper::Params pd;
    pd.rounding = clipper::Params::Rounding::DSD;
    clipper::CLIPPER cd(std::make_shared<clipper::invariants::EuclideanDistance>(iparams), pd);
    clipper::Constraint C(20, 20);
    for (int i = 0; i < 20; ++i)
      for (int j = 0; j < 20; ++j) C(i, j) = (M(i, j) != 0.0) ? 1.0 : 0.0;
    cd.setMatrixData(M, C);
    clipper::VectorXd u0(20);
    for (int i = 0; i < 20; ++i) u0(i) = 1.0;
    cd.solve(u0);
    const std::vector<int> nodes = cd.getSolution().nodes;
    EXPECT(nodes.size() >= 2);
    for (size_t i = 1; i < nodes.size(); ++i) EXPECT(nodes[i - 1] < nodes[i]);
    std::printf("DSD.Solve / SolveRestrictedGraph / Rounding::DSD ok (%zu nodes)\n", nodes.size());
  }
  std::printf("ALL FACADE TESTS PASSED\n");
  return 0;
}
