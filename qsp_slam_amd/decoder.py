"""DeepSDF decoder handle: weight loading (the only place PyTorch is touched, and only to read a checkpoint) and upload
through qsp_decoder_create.  Reference: deep_sdf/workspace.py:202-224, deep_sdf/deep_sdf_decoder.py:9-72."""
import ctypes as C
import ast
import json
import os

import numpy as np

from . import _lib


class DeepSdfDecoder(object):
    """Owns a qsp_decoder*.  `layers` is a list of (weight (out,in) f32, weight_g (out,) f32 | None, bias (out,) f32)."""

    def __init__(self, layers, latent_in=(4,), code_len=64, device=0):
        L = _lib.lib()
        latent_in = tuple(latent_in or ())
        if len(latent_in) > 1:
            raise _lib.QspError(_lib.QSP_ERR_UNSUPPORTED, "at most one latent_in layer is supported")
        n = len(layers)
        self._keep = []
        in_dim = np.array([l[0].shape[1] for l in layers], np.int32)
        out_dim = np.array([l[0].shape[0] for l in layers], np.int32)
        w = [_lib.f32c(l[0]) for l in layers]
        g = [None if l[1] is None else _lib.f32c(np.asarray(l[1]).reshape(-1)) for l in layers]
        b = [_lib.f32c(l[2]) for l in layers]
        fpp = _lib.c_float_p * n
        wp = fpp(*[_lib.fptr(a) for a in w])
        gp = fpp(*[(_lib.fptr(a) if a is not None else _lib.c_float_p()) for a in g])
        bp = fpp(*[_lib.fptr(a) for a in b])
        self._keep = [w, g, b, in_dim, out_dim]
        desc = _lib.DecoderDesc(n, int(code_len), int(latent_in[0]) if latent_in else -1, _lib.i32ptr(in_dim), _lib.i32ptr(out_dim),
                                C.cast(wp, C.POINTER(_lib.c_float_p)), C.cast(gp, C.POINTER(_lib.c_float_p)),
                                C.cast(bp, C.POINTER(_lib.c_float_p)))
        h = C.c_void_p()
        _lib.check(L.qsp_decoder_create(C.byref(desc), int(device), C.byref(h)))
        self.handle = h
        self.code_len = int(code_len)
        self.device = int(device)
        self.precision = "f32"
        self.render_screening = 0.0
        if os.environ.get("QSP_PRECISION"):        # e.g. to run a whole test session on the split-bf16 pipe
            self.set_precision(os.environ["QSP_PRECISION"])
            if self.precision == "fp16x2" and os.environ.get("QSP_SCREENING"):
                # ... or on the screened split-fp16 pipe, every batch in two passes whatever its size (QSP_SCREENING = margin)
                self.set_render_screening(float(os.environ["QSP_SCREENING"]))
                self.set_screening_min_samples(0)
                if os.environ.get("QSP_DEPTH_STAGING"):      # ... and in two depth stages whatever its size ("always"), or never ("0")
                    self.set_depth_staging({"always": "always", "0": False}.get(os.environ["QSP_DEPTH_STAGING"], True))
        self.mac_per_point = int(sum(int(i) * int(o) for i, o in zip(in_dim, out_dim)))

    PRECISIONS = {"f32": 0, "bf16x3": 1, "fp16x2": 2}

    def set_precision(self, name):
        """"f32": every multiply-add of the decoder on the exact-f32 matrix pipe (default).  "bf16x3": operands as three bf16
        terms, six products per multiply-add on the bf16 matrix pipe.  "fp16x2": operands as two fp16 terms (the second
        pre-scaled by 2^11), three products on the fp16 matrix pipe; refuses decoders whose weights leave fp16's range and
        fails the call if an activation or gradient does.  All accumulate in f32 and are float32-equivalent in accuracy
        (include/qsp_hip.h, QSP_DEC_OPT_*)."""
        if name not in self.PRECISIONS:
            raise ValueError("precision must be one of %s" % sorted(self.PRECISIONS))
        self.set_forward_precision(self.PRECISIONS[name])
        self.set_jacobian_precision(self.PRECISIONS[name])
        self.precision = name
        if name != "fp16x2":
            self.render_screening = 0.0      # (the library drops the option with the pipe it belongs to)

    def set_forward_precision(self, mode):
        """forward-only passes (decode_sdf, mesh grid, ray samples): 0 / False = exact-f32 matrix pipe (default), 1 / True =
        split bf16, 2 = split fp16; see QSP_DEC_OPT_FORWARD_PRECISION in qsp_hip.h"""
        _lib.check(_lib.lib().qsp_decoder_set_option(self.handle, 1, int(mode)))

    def set_jacobian_precision(self, mode):
        """the forward+backward pass (sdf_value_grad, the fused Jacobian / normal-equation kernel) likewise
        (QSP_DEC_OPT_JACOBIAN_PRECISION)"""
        _lib.check(_lib.lib().qsp_decoder_set_option(self.handle, 2, int(mode)))

    def set_tile_points(self, n):
        """points per MLP tile (64 default, or 32) of the refinement batches created after the call; 32 is the latency option for
        one object per call and exists on the "fp16x2" pipe only (QSP_DEC_OPT_TILE_POINTS in qsp_hip.h)"""
        _lib.check(_lib.lib().qsp_decoder_set_option(self.handle, 3, int(n)))
        self.tile_points = int(n)

    def set_screening_min_samples(self, n=-1):
        """a run is screened only when its batch holds more than n ray samples; -1 = automatic (more than two rounds of 64-point
        tiles over the chip), 0 = always (QSP_DEC_OPT_SCREENING_MIN_SAMPLES)"""
        _lib.check(_lib.lib().qsp_decoder_set_option(self.handle, 7, int(n)))

    def set_depth_staging(self, on=True):
        """the screened forward in two depth stages: samples behind a ray's first opaque sample are multiplied by an exact zero
        transmittance and are not evaluated (QSP_DEC_OPT_DEPTH_STAGING; bit-identical results).  True: large batches only (the
        default: a stage costs launches that only pay beyond ~2 M samples); "always": whatever the size (tests); False: never."""
        _lib.check(_lib.lib().qsp_decoder_set_option(self.handle, 10, 2 if on == "always" else (1 if on else 0)))

    def set_screen_audit(self, one_in=100):
        """the screened forward's out-of-band audit (QSP_DEC_OPT_SCREEN_AUDIT): one in `one_in` of the samples the screening pass
        put OUTSIDE the band is re-evaluated on the split-fp16 tile as well; one found inside the cut-off repeats the run in one
        pass.  0 = off, 1 = every sample."""
        _lib.check(_lib.lib().qsp_decoder_set_option(self.handle, 9, int(one_in)))

    def set_render_screening(self, margin=0.01):
        """two-pass ray-sample forward of the refinement on the "fp16x2" pipe (QSP_DEC_OPT_RENDER_SCREENING in qsp_hip.h): every
        sample on a one-product tile, only those with |s1| < cut_off + margin on the split-fp16 tile.  Bit-identical results to the
        unscreened "fp16x2" path as long as margin covers |s1 - s3| (profiles/r03_screen_margin.txt); 0 / None turns it off."""
        _lib.check(_lib.lib().qsp_decoder_set_option(self.handle, 4, int(round(1e6 * float(margin or 0.0)))))
        self.render_screening = float(margin or 0.0)

    def set_use_tanh(self, on):
        """NetworkSpecs.use_tanh (deep_sdf/deep_sdf_decoder.py:66-68,92-94): a tanh on the output layer in front of the final one"""
        _lib.check(_lib.lib().qsp_decoder_set_option(self.handle, 5, 1 if on else 0))
        self.use_tanh = bool(on)

    def set_range_fallback(self, on):
        """on (default): a call during which a split-fp16 kernel met a value outside fp16's range is repeated on the f32 pipe by
        the library and returns normally; off: that call raises QSP_ERR_UNSUPPORTED (QSP_DEC_OPT_RANGE_FALLBACK)"""
        _lib.check(_lib.lib().qsp_decoder_set_option(self.handle, 6, 1 if on else 0))

    @property
    def range_fallbacks(self):
        """calls of this decoder that were repeated on the f32 pipe so far"""
        return int(_lib.lib().qsp_decoder_get_counter(self.handle, 1))

    @property
    def screen_fallbacks(self):
        """runs the screening self-check repeated in one pass so far (|s1 - s3| on a band sample above half the margin)"""
        return int(_lib.lib().qsp_decoder_get_counter(self.handle, 5))

    @property
    def narrow_tile(self):
        """True when the split-fp16 kernels run their NARROW form for this decoder (much smaller than the 8 x 512 shape it is embedded
        in: identity slots, all-zero slabs and column blocks skipped -- QSP_DEC_OPT_NARROW_TILE)"""
        return bool(_lib.lib().qsp_decoder_get_counter(self.handle, 4))

    def set_narrow_tile(self, on):
        _lib.check(_lib.lib().qsp_decoder_set_option(self.handle, 8, 1 if on else 0))

    @property
    def arena_stats(self):
        """(calls of qsp_reconstruct_objects that refilled the decoder's resident batch, calls that had to (re)allocate it)"""
        L = _lib.lib()
        return int(L.qsp_decoder_get_counter(self.handle, 2)), int(L.qsp_decoder_get_counter(self.handle, 3))

    def decode_sdf_screen(self, code, x):
        """the screening tile's values (first pass of set_render_screening) on explicit points -- diagnostic, not SDF values"""
        x = _lib.f32c(x)
        code = _lib.f32c(np.asarray(code)[: self.code_len])
        out = np.empty(x.shape[0], np.float32)
        _lib.check(_lib.lib().qsp_decode_sdf_screen(self.handle, _lib.fptr(code), _lib.fptr(x), x.shape[0], _lib.fptr(out)))
        return out

    def close(self):
        if getattr(self, "handle", None):
            _lib.lib().qsp_decoder_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- constructors -------------------------------------------------------------------------------------------
    @classmethod
    def from_state_dict(cls, state, latent_in=(4,), code_len=64, device=0):
        """`state`: mapping 'lin{l}.weight_v' / '.weight_g' / '.weight' / '.bias' -> array-like ('module.' prefix of
        DataParallel checkpoints tolerated, deep_sdf/workspace.py:215-220)."""
        st = {}
        for k, v in state.items():
            k = k[len("module."):] if k.startswith("module.") else k
            st[k] = np.asarray(v.detach().cpu().numpy() if hasattr(v, "detach") else v, dtype=np.float32)
        norm = [k for k in st if k.startswith("bn")]
        if norm:       # deep_sdf_decoder.py:58-63,96-102: LayerNorm in place of weight norm -- a different function, not built
            raise _lib.QspError(_lib.QSP_ERR_UNSUPPORTED, "decoder family: LayerNorm parameters (%s ...) in the checkpoint: "
                                "norm_layers without weight_norm is not supported" % norm[0])
        layers = []
        l = 0
        while ("lin%d.bias" % l) in st:
            if ("lin%d.weight_v" % l) in st:
                layers.append((st["lin%d.weight_v" % l], st["lin%d.weight_g" % l], st["lin%d.bias" % l]))
            else:
                layers.append((st["lin%d.weight" % l], None, st["lin%d.bias" % l]))
            l += 1
        return cls(layers, latent_in=latent_in, code_len=code_len, device=device)

    @classmethod
    def from_npz(cls, path, device=0):
        z = np.load(path, allow_pickle=False)
        meta = ast.literal_eval(str(z["meta"]))     # a literal dict; never evaluate file contents as code
        if not (isinstance(meta, dict) and isinstance(meta.get("latent_size"), int)
                and all(isinstance(i, int) for i in meta.get("latent_in", ()))):
            raise ValueError("malformed decoder meta in %s" % path)
        cls.check_network_specs({k: meta[k] for k in ("xyz_in_all", "norm_layers", "weight_norm") if k in meta})
        dec = cls.from_state_dict({k: z[k] for k in z.files if k != "meta"}, latent_in=meta["latent_in"],
                                  code_len=meta["latent_size"], device=device)
        if meta.get("use_tanh", False):
            dec.set_use_tanh(True)
        return dec

    @classmethod
    def from_experiment_dir(cls, experiment_directory, checkpoint="latest", device=0):
        """specs.json + ModelParameters/<checkpoint>.pth, as deep_sdf/workspace.py:202-224 reads them."""
        specs_filename = os.path.join(experiment_directory, "specs.json")
        if not os.path.isfile(specs_filename):
            raise Exception('The experiment directory does not include specifications file "specs.json"')
        specs = json.load(open(specs_filename))
        import torch  # checkpoint format only

        saved = torch.load(os.path.join(experiment_directory, "ModelParameters", checkpoint + ".pth"),
                           map_location="cpu")
        ns = specs["NetworkSpecs"]
        cls.check_network_specs(ns)
        dec = cls.from_state_dict(saved["model_state_dict"], latent_in=tuple(ns.get("latent_in", ())),
                                  code_len=specs["CodeLength"], device=device)
        if ns.get("use_tanh", False):
            dec.set_use_tanh(True)
        return dec

    @staticmethod
    def check_network_specs(ns):
        """Every key of NetworkSpecs that the reference's Decoder constructor honours (deep_sdf/deep_sdf_decoder.py:9-72) is
        either implemented or refused: a decoder is never loaded as a different function without an error.
          dims, latent_in, weight_norm + norm_layers, use_tanh   implemented
          dropout, dropout_prob, latent_dropout                  no effect in eval mode (F.dropout(training=False) is the identity)
          xyz_in_all                                             refused (QSP_ERR_UNSUPPORTED)
          norm_layers without weight_norm (LayerNorm)            refused"""
        if ns.get("xyz_in_all", False):
            raise _lib.QspError(_lib.QSP_ERR_UNSUPPORTED, "decoder family: xyz_in_all is not supported")
        if ns.get("norm_layers") and not ns.get("weight_norm", False):
            raise _lib.QspError(_lib.QSP_ERR_UNSUPPORTED, "decoder family: norm_layers without weight_norm (LayerNorm) is not supported")
        known = {"dims", "dropout", "dropout_prob", "norm_layers", "latent_in", "weight_norm", "xyz_in_all", "use_tanh",
                 "latent_dropout"}
        unknown = sorted(set(ns) - known)
        if unknown:
            raise _lib.QspError(_lib.QSP_ERR_UNSUPPORTED, "decoder family: unknown NetworkSpecs keys %s" % unknown)

    # ---- entry points of reconstruct/loss_utils.py ----------------------------------------------------------------
    def decode_sdf(self, code, x):
        """decode_sdf(decoder, lat_vec, x), loss_utils.py:51-79 -> (N,) float32"""
        x = _lib.f32c(x)
        code = _lib.f32c(np.asarray(code)[: self.code_len])
        out = np.empty(x.shape[0], np.float32)
        _lib.check(_lib.lib().qsp_decode_sdf(self.handle, _lib.fptr(code), _lib.fptr(x), x.shape[0], _lib.fptr(out)))
        return out

    def sdf_value_grad(self, code, x):
        """get_batch_sdf_jacobian(decoder, lat_vec, x, 1), loss_utils.py:82-103 -> y (N,), grad (N, code_len+3)"""
        x = _lib.f32c(x)
        code = _lib.f32c(np.asarray(code)[: self.code_len])
        y = np.empty(x.shape[0], np.float32)
        g = np.empty((x.shape[0], self.code_len + 3), np.float32)
        _lib.check(_lib.lib().qsp_sdf_value_grad(self.handle, _lib.fptr(code), _lib.fptr(x), x.shape[0], _lib.fptr(y),
                                                 _lib.fptr(g)))
        return y, g
