// reconstruct_hip -- pybind11 module with the reference's B1 entry points as C++ host code over the C-ABI
// of include/qsp_hip.h: the same Optimizer / MeshExtractor names, constructor arguments, method signatures and result object
// as reconstruct/optimizer.py:26-304 of the reference, for embedders that want no ctypes between pybind11::embed and the
// library.  (qsp_slam_amd/reconstruct/optimizer.py is the ctypes twin; tests/test_gpu_pybind.py checks that both give the
// same bits.)  No torch, no Eigen: numpy arrays in any stride order are accepted (pybind11 hands Eigen::MatrixXf over as
// Fortran-ordered arrays, src/LocalMapping_util.cc:705-706) and copied to contiguous float32.
//
// The decoder handle is the one qsp_slam_amd.DeepSdfDecoder owns (weights are loaded and folded there): `decoder.handle`.
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>

#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/qsp_hip.h"

namespace py = pybind11;
using farr = py::array_t<float, py::array::c_style | py::array::forcecast>;

static void check(int rc) {
    if (rc != QSP_OK) throw std::runtime_error("qsp_hip error " + std::to_string(rc) + ": " + qsp_last_error());
}

static qsp_decoder* handle_of(const py::object& decoder) {
    // ctypes.c_void_p -> integer address
    py::object h = decoder.attr("handle");
    py::object v = py::hasattr(h, "value") ? h.attr("value") : h;
    if (v.is_none()) throw std::runtime_error("decoder is closed");
    return reinterpret_cast<qsp_decoder*>(v.cast<uintptr_t>());
}

static py::object result_object(py::object t_cam_obj, py::object code, bool good, float loss) {
    // the reference returns ForceKeyErrorDict(t_cam_obj=..., code=..., is_good=..., loss=...)  (optimizer.py:276-281)
    py::object cls = py::module_::import("qsp_slam_amd.reconstruct.utils").attr("ForceKeyErrorDict");
    py::dict kw;
    kw["t_cam_obj"] = t_cam_obj;
    kw["code"] = code;
    kw["is_good"] = good;
    kw["loss"] = loss;
    return cls(**kw);
}

struct Optimizer {
    py::object decoder;      // keeps the Python owner of the handle alive
    qsp_joint_cfg cfg{};
    int code_len = 64;
    int n_iter_pose_only = 5;
    bool debug = false;

    Optimizer(py::object decoder_, py::object configs, bool debug_) : decoder(decoder_), debug(debug_) {
        py::object oc = configs.attr("optimizer");          // missing keys raise KeyError, as in the reference
        py::object jo = oc.attr("joint_optim");
        cfg.k1 = jo.attr("k1").cast<float>();
        cfg.k2 = jo.attr("k2").cast<float>();
        cfg.k3 = jo.attr("k3").cast<float>();
        cfg.k4 = jo.attr("k4").cast<float>();
        cfg.b1 = jo.attr("b1").cast<float>();
        cfg.b2 = jo.attr("b2").cast<float>();
        cfg.lr = jo.attr("learning_rate").cast<float>();
        cfg.s_damp = jo.attr("scale_damping").cast<float>();
        cfg.n_iter = jo.attr("num_iterations").cast<int>();
        code_len = oc.attr("code_len").cast<int>();
        cfg.code_len = code_len;
        cfg.n_depth = oc.attr("num_depth_samples").cast<int>();
        cfg.cut_off = oc.attr("cut_off_threshold").cast<float>();
        if (configs.attr("data_type").cast<std::string>() == "KITTI")
            n_iter_pose_only = oc.attr("pose_only_optim").attr("num_iterations").cast<int>();
    }

    py::object reconstruct_object(farr t_cam_obj, farr pts, farr rays, farr depth, py::object code) {
        if (t_cam_obj.size() != 16) throw std::invalid_argument("t_cam_obj must be (4,4)");
        if (pts.size() % 3 || rays.size() % 3) throw std::invalid_argument("pts and rays must be (n,3)");
        const int32_t n_pts = (int32_t)(pts.size() / 3), n_rays = (int32_t)(rays.size() / 3), n_fg = (int32_t)depth.size();
        const float* pp = pts.data();
        const float* rp = rays.data();
        const float* dp = depth.data();
        const int32_t hyp = 0;
        std::vector<float> c0;
        if (!code.is_none()) {
            farr c = code.cast<farr>();
            if ((int)c.size() < code_len) throw std::invalid_argument("code shorter than code_len");
            c0.assign(c.data(), c.data() + code_len);          // code[:code_len]  (optimizer.py:118)
        }
        py::array_t<float> T({4, 4}), cd(code_len);
        float loss = 0.f;
        uint8_t good = 0;
        qsp_decoder* h = handle_of_cached();
        {
            py::gil_scoped_release nogil;
            check(qsp_reconstruct_objects(h, &cfg, 1, &pp, &n_pts, &rp, &n_rays, &dp, &n_fg, 1, &hyp,
                                          t_cam_obj.data(), c0.empty() ? nullptr : c0.data(), T.mutable_data(),
                                          cd.mutable_data(), &loss, &good));
        }
        if (!good) return result_object(py::none(), py::none(), false, loss);
        return result_object(T, cd, true, loss);
    }

    py::object estimate_pose_cam_obj(farr t_co_se3, float scale, farr pts, farr code) {
        if (t_co_se3.size() != 16) throw std::invalid_argument("t_co_se3 must be (4,4)");
        if ((int)code.size() < code_len) throw std::invalid_argument("code shorter than code_len");
        const int32_t n_pts = (int32_t)(pts.size() / 3);
        const float* pp = pts.data();
        py::array_t<float> T({4, 4});
        qsp_decoder* h = handle_of_cached();
        {
            py::gil_scoped_release nogil;
            check(qsp_estimate_pose(h, 1, t_co_se3.data(), &scale, &pp, &n_pts, code.data(), n_iter_pose_only,
                                    T.mutable_data()));
        }
        return T;
    }

    qsp_decoder* handle_of_cached() { return h_ ? h_ : (h_ = handle_of(decoder)); }
    qsp_decoder* h_ = nullptr;
};

struct MeshExtractor {
    py::object decoder;
    qsp_mesh_extractor* m = nullptr;
    int code_len, voxels_dim;

    MeshExtractor(py::object decoder_, int code_len_, int voxels_dim_) : decoder(decoder_), code_len(code_len_), voxels_dim(voxels_dim_) {
        // create_voxel_grid of the Python twin reproduces the reference's true-division quirk (reconstruct/utils.py:98-117)
        farr grid = py::module_::import("qsp_slam_amd.reconstruct.optimizer").attr("create_voxel_grid")(voxels_dim).cast<farr>();
        check(qsp_mesh_extractor_create(handle_of(decoder), voxels_dim, grid.data(), &m));
    }
    ~MeshExtractor() {
        if (m) qsp_mesh_extractor_destroy(m);
    }
    py::object extract_mesh_from_code(farr code) {
        if ((int)code.size() < code_len) throw std::invalid_argument("code shorter than code_len");
        int64_t nv = 0, nf = 0;
        check(qsp_mesh_extract(m, code.data(), &nv, &nf));
        if (nv == 0) throw std::runtime_error("No surface found at the given iso value.");      // (as skimage's marching_cubes_lewiner)
        py::array_t<double> verts({(py::ssize_t)nv, (py::ssize_t)3});       // float64, as convert_sdf_voxels_to_mesh returns them
        py::array_t<int32_t> faces({(py::ssize_t)nf, (py::ssize_t)3});
        check(qsp_mesh_fetch(m, nullptr, faces.mutable_data(), nullptr));
        check(qsp_mesh_fetch_f64(m, verts.mutable_data()));
        py::object cls = py::module_::import("qsp_slam_amd.reconstruct.utils").attr("ForceKeyErrorDict");
        py::dict kw;
        kw["vertices"] = verts;
        kw["faces"] = faces;
        return cls(**kw);
    }
};

PYBIND11_MODULE(reconstruct_hip, mod) {
    mod.doc() = "QSP-SLAM reconstruct.optimizer entry points over libqsp_hip.so (C++ host, pybind11)";
    mod.def("version", []() { return qsp_version(); });
    mod.def("device_count", []() { return qsp_device_count(); });
    py::class_<Optimizer>(mod, "Optimizer")
        .def(py::init<py::object, py::object, bool>(), py::arg("decoder"), py::arg("configs"), py::arg("debug") = false)
        .def_readonly("code_len", &Optimizer::code_len)
        .def("reconstruct_object", &Optimizer::reconstruct_object, py::arg("t_cam_obj"), py::arg("pts"), py::arg("rays"),
             py::arg("depth"), py::arg("code") = py::none())
        .def("estimate_pose_cam_obj", &Optimizer::estimate_pose_cam_obj, py::arg("t_co_se3"), py::arg("scale"), py::arg("pts"),
             py::arg("code"));
    py::class_<MeshExtractor>(mod, "MeshExtractor")
        .def(py::init<py::object, int, int>(), py::arg("decoder"), py::arg("code_len") = 64, py::arg("voxels_dim") = 64)
        .def("extract_mesh_from_code", &MeshExtractor::extract_mesh_from_code, py::arg("code"));
}
