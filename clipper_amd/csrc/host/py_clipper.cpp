// py_clipper.cpp — pybind11 module `clipperpy`: the reference's Python surface
// (bindings/python/py_clipper.cpp:116-233, trampolines.h:14-30) over the clipper:: facade of
// this build. Same module / submodule / class / method / field names and argument order.
//
// The reference converts numpy <-> Eigen with pybind11/eigen.h and `noconvert()` arguments
// (float64 data, int32 associations). Eigen is not available here, so small type casters do
// the same for the facade's containers: dtype must match when the argument is `noconvert`,
// any memory layout is accepted and copied to column-major.
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <cstring>
#include <sstream>

#include "clipper/clipper.h"
#include "clipper/utils.h"

namespace py = pybind11;
using namespace pybind11::literals;

#ifndef CLIPPER_VERSION
#define CLIPPER_VERSION "0.2.4+mi355x.1"
#endif

// ------------------------------------------------------------------------------------------
// numpy <-> facade containers
// ------------------------------------------------------------------------------------------
namespace pybind11 {
namespace detail {

template <typename T>
struct dense_matrix_caster {
  using Mat = clipper::DenseMatrix<T>;
  static bool load_into(handle src, bool convert, Mat& out, int fixed_cols) {
    if (!src) return false;
    if (!convert && !isinstance<array_t<T>>(src)) return false;  // `noconvert`: dtype must match
    auto arr = array_t<T, array::f_style | array::forcecast>::ensure(src);
    if (!arr) return false;
    if (arr.ndim() == 1 && arr.shape(0) == 0) {  // empty -> 0 x cols
      out.resize(0, fixed_cols > 0 ? fixed_cols : 0);
      return true;
    }
    if (arr.ndim() != 2) return false;
    if (fixed_cols > 0 && arr.shape(1) != fixed_cols && arr.size() != 0) return false;
    out.resize(arr.shape(0), arr.shape(1));
    if (arr.size() > 0) std::memcpy(out.data(), arr.data(), sizeof(T) * static_cast<size_t>(arr.size()));
    return true;
  }
  static handle to_numpy(const Mat& m) {
    array_t<T, array::f_style> a({m.rows(), m.cols()});
    if (m.size() > 0) std::memcpy(a.mutable_data(), m.data(), sizeof(T) * static_cast<size_t>(m.size()));
    return a.release();
  }
};

template <>
struct type_caster<clipper::MatrixXd> {
  PYBIND11_TYPE_CASTER(clipper::MatrixXd, const_name("numpy.ndarray[numpy.float64[m, n]]"));
  bool load(handle src, bool convert) {
    return dense_matrix_caster<double>::load_into(src, convert, value, 0);
  }
  static handle cast(const clipper::MatrixXd& m, return_value_policy, handle) {
    return dense_matrix_caster<double>::to_numpy(m);
  }
};

template <>
struct type_caster<clipper::Association> {
  PYBIND11_TYPE_CASTER(clipper::Association, const_name("numpy.ndarray[numpy.int32[m, 2]]"));
  bool load(handle src, bool convert) {
    return dense_matrix_caster<int>::load_into(src, convert, value, 2);
  }
  static handle cast(const clipper::Association& m, return_value_policy, handle) {
    return dense_matrix_caster<int>::to_numpy(m);
  }
};

template <>
struct type_caster<clipper::VectorXd> {
  PYBIND11_TYPE_CASTER(clipper::VectorXd, const_name("numpy.ndarray[numpy.float64[m, 1]]"));
  bool load(handle src, bool convert) {
    if (!src) return false;
    if (!convert && !isinstance<array_t<double>>(src)) return false;
    auto arr = array_t<double, array::c_style | array::forcecast>::ensure(src);
    if (!arr) return false;
    if (arr.ndim() > 2) return false;
    if (arr.ndim() == 2 && arr.shape(0) != 1 && arr.shape(1) != 1 && arr.size() != 0) return false;
    value.resize(arr.size());
    if (arr.size() > 0) std::memcpy(value.data(), arr.data(), sizeof(double) * static_cast<size_t>(arr.size()));
    return true;
  }
  static handle cast(const clipper::VectorXd& v, return_value_policy, handle) {
    array_t<double> a(static_cast<py::ssize_t>(v.size()));
    if (v.size() > 0) std::memcpy(a.mutable_data(), v.data(), sizeof(double) * static_cast<size_t>(v.size()));
    return a.release();
  }
};

}  // namespace detail
}  // namespace pybind11

// ------------------------------------------------------------------------------------------
// trampolines (reference bindings/python/trampolines.h:14-30)
// ------------------------------------------------------------------------------------------
template <class InvariantBase = clipper::invariants::Invariant>
class PyInvariant : public InvariantBase {
 public:
  using InvariantBase::InvariantBase;
};

template <class PairwiseInvariantBase = clipper::invariants::PairwiseInvariant>
class PyPairwiseInvariant : public PyInvariant<PairwiseInvariantBase> {
 public:
  using PyInvariant<PairwiseInvariantBase>::PyInvariant;
  using Datum = clipper::invariants::Datum;
  double operator()(const Datum& ai, const Datum& aj, const Datum& bi, const Datum& bj) override {
    pybind11::gil_scoped_acquire acquire;  // the scoring loop may run without the GIL
    PYBIND11_OVERRIDE_PURE_NAME(double, PairwiseInvariantBase, "__call__", operator(), ai, aj, bi, bj);
  }
};

// non-pure variant for the built-ins (a Python subclass may or may not override __call__)
template <class Builtin>
class PyBuiltinInvariant : public Builtin {
 public:
  using Builtin::Builtin;
  using Datum = clipper::invariants::Datum;
  double operator()(const Datum& ai, const Datum& aj, const Datum& bi, const Datum& bj) override {
    pybind11::gil_scoped_acquire acquire;
    PYBIND11_OVERRIDE_NAME(double, Builtin, "__call__", operator(), ai, aj, bi, bj);
  }
};

// ------------------------------------------------------------------------------------------

void pybind_invariants(py::module& m) {
  m.doc() =
      "Invariants are quantities that do not change under the transformation between two sets "
      "of objects. They are used to build a consistency graph. Some built-in invariants are "
      "provided.";
  using namespace clipper::invariants;

  py::class_<Invariant, PyInvariant<>, std::shared_ptr<Invariant>>(m, "Invariant").def(py::init<>());
  py::class_<PairwiseInvariant, Invariant, PyPairwiseInvariant<>, std::shared_ptr<PairwiseInvariant>>(
      m, "PairwiseInvariant")
      .def(py::init<>())
      .def("__call__", &PairwiseInvariant::operator());

  py::class_<EuclideanDistance::Params>(m, "EuclideanDistanceParams")
      .def(py::init<>())
      .def("__repr__",
           [](const EuclideanDistance::Params& p) {
             std::ostringstream r;
             r << "<EuclideanDistanceParams : sigma=" << p.sigma << " epsilon=" << p.epsilon
               << " mindist=" << p.mindist << ">";
             return r.str();
           })
      .def_readwrite("sigma", &EuclideanDistance::Params::sigma)
      .def_readwrite("epsilon", &EuclideanDistance::Params::epsilon)
      .def_readwrite("mindist", &EuclideanDistance::Params::mindist);

  py::class_<EuclideanDistance, PairwiseInvariant, PyBuiltinInvariant<EuclideanDistance>,
             std::shared_ptr<EuclideanDistance>>(m, "EuclideanDistance")
      .def(py::init<const EuclideanDistance::Params&>());

  py::class_<PointNormalDistance::Params>(m, "PointNormalDistanceParams")
      .def(py::init<>())
      .def("__repr__",
           [](const PointNormalDistance::Params& p) {
             std::ostringstream r;
             r << "<PointNormalDistanceParams : sigp=" << p.sigp << " epsp=" << p.epsp
               << " sign=" << p.sign << " epsn=" << p.epsn << ">";
             return r.str();
           })
      .def_readwrite("sigp", &PointNormalDistance::Params::sigp)
      .def_readwrite("epsp", &PointNormalDistance::Params::epsp)
      .def_readwrite("sign", &PointNormalDistance::Params::sign)
      .def_readwrite("epsn", &PointNormalDistance::Params::epsn);

  py::class_<PointNormalDistance, PairwiseInvariant, PyBuiltinInvariant<PointNormalDistance>,
             std::shared_ptr<PointNormalDistance>>(m, "PointNormalDistance")
      .def(py::init<const PointNormalDistance::Params&>());
}

void pybind_utils(py::module& m) {
  m.doc() = "Various convenience utilities for working with CLIPPER";
  m.def("create_all_to_all", clipper::utils::createAllToAll, "n1"_a, "n2"_a,
        "Create an all-to-all hypothesis for association. Useful for the case of no prior "
        "information or putative associations.");
  m.def("k2ij", clipper::utils::k2ij, "k"_a, "n"_a,
        "Maps a flat index k to coordinate of a square nxn symmetric matrix");
}

PYBIND11_MODULE(clipperpy, m) {
  m.doc() = "A graph-theoretic framework for robust data association (MI355X hot path)";
  m.attr("__version__") = CLIPPER_VERSION;

  py::module m_invariants = m.def_submodule("invariants");
  pybind_invariants(m_invariants);

  py::module m_utils = m.def_submodule("utils");
  pybind_utils(m_utils);

  // The reference fills `clipperpy.dsd` with pybind_utils (py_clipper.cpp:127-128; pybind_dsd is
  // never called), so that is what existing scripts see; the exact DSD solver is out of scope.
  py::module m_dsd = m.def_submodule("dsd");
  pybind_utils(m_dsd);

  py::enum_<clipper::maxclique::Method>(m, "MCMethod")
      .value("EXACT", clipper::maxclique::Method::EXACT)
      .value("HEU", clipper::maxclique::Method::HEU)
      .value("KCORE", clipper::maxclique::Method::KCORE);

  py::class_<clipper::maxclique::Params>(m, "MCParams")
      .def(py::init<>())
      .def("__repr__", [](const clipper::maxclique::Params&) { return "<CLIPPER Maximum Clique Parameters>"; })
      .def_readwrite("method", &clipper::maxclique::Params::method)
      .def_readwrite("threads", &clipper::maxclique::Params::threads)
      .def_readwrite("time_limit", &clipper::maxclique::Params::time_limit)
      .def_readwrite("verbose", &clipper::maxclique::Params::verbose);

  py::class_<clipper::sdp::Params>(m, "SDPParams")
      .def(py::init<>())
      .def("__repr__", [](const clipper::sdp::Params&) { return "<CLIPPER SDP Parameters>"; })
      .def_readwrite("verbose", &clipper::sdp::Params::verbose)
      .def_readwrite("max_iters", &clipper::sdp::Params::max_iters)
      .def_readwrite("acceleration_interval", &clipper::sdp::Params::acceleration_interval)
      .def_readwrite("acceleration_lookback", &clipper::sdp::Params::acceleration_lookback)
      .def_readwrite("eps_abs", &clipper::sdp::Params::eps_abs)
      .def_readwrite("eps_rel", &clipper::sdp::Params::eps_rel)
      .def_readwrite("eps_infeas", &clipper::sdp::Params::eps_infeas)
      .def_readwrite("time_limit_secs", &clipper::sdp::Params::time_limit_secs);

  py::enum_<clipper::Params::Rounding>(m, "Rounding")
      .value("NONZERO", clipper::Params::Rounding::NONZERO)
      .value("DSD", clipper::Params::Rounding::DSD)
      .value("DSD_HEU", clipper::Params::Rounding::DSD_HEU)
      .export_values();

  py::class_<clipper::Params>(m, "Params")
      .def(py::init<>())
      .def("__repr__", [](const clipper::Params&) { return "<CLIPPER Parameters>"; })
      .def_readwrite("tol_u", &clipper::Params::tol_u)
      .def_readwrite("tol_F", &clipper::Params::tol_F)
      .def_readwrite("tol_Fop", &clipper::Params::tol_Fop)
      .def_readwrite("maxiniters", &clipper::Params::maxiniters)
      .def_readwrite("maxoliters", &clipper::Params::maxoliters)
      .def_readwrite("beta", &clipper::Params::beta)
      .def_readwrite("maxlsiters", &clipper::Params::maxlsiters)
      .def_readwrite("eps", &clipper::Params::eps)
      .def_readwrite("affinityeps", &clipper::Params::affinityeps)
      .def_readwrite("rescale_u0", &clipper::Params::rescale_u0)
      .def_readwrite("rounding", &clipper::Params::rounding);

  py::class_<clipper::Solution>(m, "Solution")
      .def(py::init<>())
      .def("__repr__", [](const clipper::Solution&) { return "<CLIPPER Solution>"; })
      .def_readwrite("t", &clipper::Solution::t)
      .def_readwrite("ifinal", &clipper::Solution::ifinal)
      .def_readwrite("nodes", &clipper::Solution::nodes)
      .def_readwrite("u0", &clipper::Solution::u0)
      .def_readwrite("u", &clipper::Solution::u)
      .def_readwrite("score", &clipper::Solution::score);

  py::enum_<clipper::CLIPPER::Storage>(m, "Storage")
      .value("F32", clipper::CLIPPER::Storage::F32)
      .value("F64", clipper::CLIPPER::Storage::F64)
      .value("F32_CSC", clipper::CLIPPER::Storage::F32_CSC)
      .value("F64_CSC", clipper::CLIPPER::Storage::F64_CSC);

  py::class_<clipper::CLIPPER>(m, "CLIPPER")
      .def(py::init([](const clipper::invariants::PairwiseInvariantPtr& invariant,
                       const clipper::Params& params) {
        clipper::CLIPPER* c = new clipper::CLIPPER(invariant, params);
        // Python-extended invariants cannot be evaluated from several OpenMP threads
        // (reference py_clipper.cpp:199-209, pybind11 issue 813): a Python subclass is an
        // instance of one of the trampoline types.
        const bool python_subclass =
            static_cast<bool>(std::dynamic_pointer_cast<PyPairwiseInvariant<>>(invariant)) ||
            static_cast<bool>(std::dynamic_pointer_cast<
                              PyBuiltinInvariant<clipper::invariants::EuclideanDistance>>(invariant)) ||
            static_cast<bool>(std::dynamic_pointer_cast<
                              PyBuiltinInvariant<clipper::invariants::PointNormalDistance>>(invariant));
        c->setParallelize(!python_subclass);
        return c;
      }),
           // keep the Python invariant object (and with it a Python-side __call__ override) alive
           // for as long as the CLIPPER object holds its C++ half
           py::keep_alive<1, 2>())
      .def("__repr__", [](const clipper::CLIPPER&) { return "<CLIPPER>"; })
      .def("score_pairwise_consistency", &clipper::CLIPPER::scorePairwiseConsistency,
           "D1"_a.noconvert(), "D2"_a.noconvert(), "A"_a.noconvert())
      .def("solve", &clipper::CLIPPER::solve, "u0"_a.noconvert() = clipper::VectorXd())
      .def("solve_as_maximum_clique", &clipper::CLIPPER::solveAsMaximumClique,
           "params"_a = clipper::maxclique::Params{})
      .def("solve_as_msrc_sdr", &clipper::CLIPPER::solveAsMSRCSDR, "params"_a = clipper::sdp::Params{})
      .def("get_initial_associations", &clipper::CLIPPER::getInitialAssociations)
      .def("get_selected_associations", &clipper::CLIPPER::getSelectedAssociations)
      .def("get_solution", &clipper::CLIPPER::getSolution)
      .def("get_affinity_matrix", &clipper::CLIPPER::getAffinityMatrix)
      .def("get_constraint_matrix", &clipper::CLIPPER::getConstraintMatrix)
      .def("set_matrix_data", &clipper::CLIPPER::setMatrixData, "M"_a.noconvert(), "C"_a.noconvert())
      .def("set_parallelize", &clipper::CLIPPER::setParallelize)
      // additions of this build
      .def("set_device", &clipper::CLIPPER::setDevice, "device"_a)
      .def("set_devices", &clipper::CLIPPER::setDevices, "devices"_a)
      .def("set_live_subproblem", &clipper::CLIPPER::setLiveSubproblem, "on"_a)
      .def("last_solve_passes_on_the_subproblem", &clipper::CLIPPER::lastSolvePassesOnTheSubproblem)
      .def("set_storage", &clipper::CLIPPER::setStorage, "storage"_a)
      .def("set_resident_solver", &clipper::CLIPPER::setResidentSolver, "on"_a)
      .def("set_row_views", &clipper::CLIPPER::setRowViews, "on"_a)
      .def("last_solve_passes_on_a_view", &clipper::CLIPPER::lastSolvePassesOnAView)
      .def("last_solve_was_resident", &clipper::CLIPPER::lastSolveWasResident)
      .def("get_path_stats", [](const clipper::CLIPPER& c) {
        const auto s = c.getPathStats();
        return py::dict("n_passes"_a = s.n_passes, "n_trials"_a = s.n_trials,
                        "affinity_kernel_ms"_a = s.affinity_kernel_ms, "d"_a = s.d);
      });
}
