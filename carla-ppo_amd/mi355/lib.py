"""ctypes binding of libmi355_carla.so, generated from include/mi355_carla.h.

There is NO fallback: if the shared library is missing the import of anything that needs it raises
(the product path must fail loudly without the HIP extension).
"""
import ctypes
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
HEADER = os.path.join(ROOT, "include", "mi355_carla.h")
LIB_PATH = os.environ.get("MI355_LIB") or os.path.join(HERE, "libmi355_carla.so")     # MI355_LIB: another build of the same library (tools/asan_host_check.sh)

MI_F32, MI_BF16, MI_BF16X3 = 0, 1, 2              # MI_BF16X3: split storage (two bf16 halves hi | lo per 4-byte element), include/mi355_carla.h

_CTYPES = {
    "void*": ctypes.c_void_p, "const void*": ctypes.c_void_p,
    "float*": ctypes.c_void_p, "const float*": ctypes.c_void_p,
    "double*": ctypes.c_void_p, "const double*": ctypes.c_void_p,
    "int*": ctypes.c_void_p, "const int*": ctypes.c_void_p,
    "char*": ctypes.c_char_p, "const char*": ctypes.c_char_p, "const unsigned char*": ctypes.c_void_p,
    "int": ctypes.c_int, "unsigned int": ctypes.c_uint, "float": ctypes.c_float, "double": ctypes.c_double, "long long": ctypes.c_longlong,
    "long long*": ctypes.c_void_p, "const long long*": ctypes.c_void_p, "void": None,
    "unsigned long long": ctypes.c_ulonglong, "unsigned long long*": ctypes.c_void_p,
    "void**": ctypes.c_void_p, "unsigned char*": ctypes.c_void_p, "void* const*": ctypes.c_void_p, "void*const*": ctypes.c_void_p,
    "const MiVaeDesc*": ctypes.c_void_p, "const MiPpoDesc*": ctypes.c_void_p, "const MiMlpVaeDesc*": ctypes.c_void_p,
}


class MiVaeDesc(ctypes.Structure):
    _fields_ = [("dtype", ctypes.c_int), ("max_batch", ctypes.c_int), ("ih", ctypes.c_int), ("iw", ctypes.c_int),
                ("cin", ctypes.c_int), ("ct", ctypes.c_int), ("z_dim", ctypes.c_int), ("loss_kind", ctypes.c_int),
                ("beta", ctypes.c_float), ("kl_tolerance", ctypes.c_float), ("inference_only", ctypes.c_int)]


MI_MLP_MAX_HIDDEN = 4


class MiMlpVaeDesc(ctypes.Structure):
    _fields_ = [("dtype", ctypes.c_int), ("max_batch", ctypes.c_int), ("source_size", ctypes.c_int), ("target_size", ctypes.c_int), ("z_dim", ctypes.c_int),
                ("n_enc", ctypes.c_int), ("n_dec", ctypes.c_int), ("enc", ctypes.c_int * MI_MLP_MAX_HIDDEN), ("dec", ctypes.c_int * MI_MLP_MAX_HIDDEN),
                ("loss_kind", ctypes.c_int), ("with_optimizer", ctypes.c_int), ("beta", ctypes.c_float), ("kl_tolerance", ctypes.c_float)]


class MiPpoDesc(ctypes.Structure):
    _fields_ = [("max_batch", ctypes.c_int), ("input_dim", ctypes.c_int), ("num_actions", ctypes.c_int),
                ("h1", ctypes.c_int), ("h2", ctypes.c_int), ("clip_eps", ctypes.c_float),
                ("value_scale", ctypes.c_float), ("entropy_scale", ctypes.c_float)]


class MiError(RuntimeError):
    pass


def parse_header(path=HEADER):
    """-> {name: (restype_str, [(argtype_str, argname), ...])} for every prototype in the header."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    protos = {}
    for m in re.finditer(r"^\s*((?:const\s+)?\w[\w\s]*?\**)\s*(mi_\w+)\s*\(([^)]*)\)\s*;", text, flags=re.M):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        ret = re.sub(r"\s*\*", "*", ret)
        arglist = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                mm = re.match(r"^(.*?)(\w+)$", a)
                typ = re.sub(r"\s*\*\s*", "*", mm.group(1).strip())
                arglist.append((typ, mm.group(2)))
        protos[name] = (ret, arglist)
    return protos


class _Lib:
    def __init__(self):
        if not os.path.exists(LIB_PATH):
            raise MiError("libmi355_carla.so not found at %s — run `python __graft_entry__.py` (build()) first; "
                          "there is no CPU fallback" % LIB_PATH)
        # PyTorch first: it bundles its own libamdhip64.so.7 / libhsa-runtime64 and every device buffer this library is handed comes from it.  Loaded
        # the other way round, the loader binds BOTH to the system ROCm's runtime (same SONAME) and PyTorch's build then finds no device.
        import torch  # noqa: F401
        self.cdll = ctypes.CDLL(LIB_PATH)
        self.protos = parse_header()
        self.cdll.mi_last_error.restype = ctypes.c_char_p
        self._check_structs = True
        for name, (ret, args) in self.protos.items():
            fn = getattr(self.cdll, name)          # AttributeError if the header declares a symbol the .so lacks
            fn.restype = _CTYPES[ret]
            fn.argtypes = [_CTYPES[t] for t, _ in args]
            if ret == "int" and not name.endswith(("_version", "_chunks", "_blocks", "_floats", "_bytes", "_count", "_size", "_tuning", "_ok", "_offset", "_rsag", "_recorded")):
                setattr(self, name, self._checked(name, fn))
            else:
                setattr(self, name, fn)

    def check_struct_sizes(self):
        assert self.mi_vae_desc_size() == ctypes.sizeof(MiVaeDesc), "MiVaeDesc layout mismatch between header and binding"
        assert self.mi_ppo_desc_size() == ctypes.sizeof(MiPpoDesc), "MiPpoDesc layout mismatch between header and binding"
        assert self.mi_mlpvae_desc_size() == ctypes.sizeof(MiMlpVaeDesc), "MiMlpVaeDesc layout mismatch between header and binding"

    def _checked(self, name, fn):
        def call(*a):
            rc = fn(*a)
            if rc != 0:
                raise MiError("%s failed (%d): %s" % (name, rc, self.cdll.mi_last_error().decode()))
            return rc
        call.__name__ = name
        return call


_lib = None


def get():
    global _lib
    if _lib is None:
        _lib = _Lib()
        _lib.check_struct_sizes()
    return _lib


def ptr(t):
    """Device (or host) pointer of a torch tensor / None -> int for ctypes."""
    return None if t is None else t.data_ptr()
