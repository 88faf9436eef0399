"""ctypes binding of libptnn.so (include/ptnn.h).  No fallback: a missing library or device raises PtnnError."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ABI_VERSION = 4
TASK_REG, TASK_CLS = 0, 1


class PtnnError(RuntimeError):
    pass


class Config(C.Structure):
    _fields_ = [
        ("struct_bytes", C.c_int32), ("device_id", C.c_int32), ("task", C.c_int32),
        ("n_in", C.c_int32), ("n_hidden", C.c_int32), ("n_out", C.c_int32),
        ("n_replicas_local", C.c_int32), ("n_replicas_global", C.c_int32), ("first_global_replica", C.c_int32),
        ("n_samples", C.c_int32), ("swap_interval", C.c_int32), ("pt_switch_step", C.c_int32),
        ("use_langevin", C.c_int32), ("waves_per_replica", C.c_int32), ("schedule", C.c_int32), ("groups_per_replica", C.c_int32), ("trace_capacity", C.c_int32), ("forward_bf16", C.c_int32), ("swap_rule", C.c_int32), ("shared_noise", C.c_int32), ("label_swap", C.c_int32), ("shared_device", C.c_int32), ("reserved_", C.c_int32),
        ("l_prob", C.c_float), ("learn_rate", C.c_float), ("step_w", C.c_float), ("step_eta", C.c_float),
        ("sigma_squared", C.c_float), ("nu_1", C.c_float), ("nu_2", C.c_float),
        ("seed", C.c_uint64),
    ]


def library_path():
    return os.environ.get("PTNN_LIBRARY", os.path.join(_HERE, "libptnn.so"))


_lib = None

_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int32)
_vpp = C.POINTER(C.c_void_p)

# callbacks of the host-staged transport (ptnn_all_gather_fn / ptnn_send_recv_fn of include/ptnn.h)
ALL_GATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64)
SEND_RECV_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, _ip, _ip, _vpp, C.c_int64)
XCHG_AUTO, XCHG_GATHER, XCHG_BOUNDARY = 0, 1, 2
UNIQUE_ID_BYTES = 128

# every symbol include/ptnn.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "ptnn_abi_version": (C.c_int, []),
    "ptnn_last_error": (C.c_char_p, []),
    "ptnn_supports": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "ptnn_create": (C.c_int, [C.POINTER(Config), _vpp]),
    "ptnn_destroy": (C.c_int, [C.c_void_p]),
    "ptnn_set_data": (C.c_int, [C.c_void_p, _fp, C.c_int, _fp, C.c_int, C.c_int]),
    "ptnn_set_state": (C.c_int, [C.c_void_p, _fp, _fp]),
    "ptnn_set_ladder": (C.c_int, [C.c_void_p, _fp]),
    "ptnn_run": (C.c_int, [C.c_void_p, C.c_int]),
    "ptnn_sync": (C.c_int, [C.c_void_p]),
    "ptnn_steps_done": (C.c_int, [C.c_void_p]),
    "ptnn_comm_unique_id": (C.c_int, [C.c_void_p, C.c_int]),
    "ptnn_comm_init": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "ptnn_comm_probe": (C.c_int, [_ip, C.c_int, C.POINTER(C.c_double)]),
    "ptnn_comm_info": (C.c_int, [C.c_void_p, _ip, _ip, _ip, _ip]),
    "ptnn_comm_last_stage": (C.c_int, [C.c_char_p, C.c_int]),
    "ptnn_comm_init_host": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ptnn_comm_set_mode": (C.c_int, [C.c_void_p, C.c_int]),
    "ptnn_comm_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64), _ip]),
    "ptnn_comm_finalize": (C.c_int, [C.c_void_p]),
    "ptnn_route": (C.c_int, [_ip, C.c_int, C.c_int, C.c_int, _ip, C.c_int]),
    "ptnn_run_segment": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "ptnn_swap_L_ptr": (C.c_int, [C.c_void_p, C.c_int, _vpp]),
    "ptnn_swap_set_L": (C.c_int, [C.c_void_p, C.c_int, _fp]),
    "ptnn_swap_cascade": (C.c_int, [C.c_void_p, C.c_int, _ip]),
    "ptnn_swap_row_ptr": (C.c_int, [C.c_void_p, C.c_int, _vpp, _vpp]),
    "ptnn_state_row_floats": (C.c_int, [C.c_void_p]),
    "ptnn_stream": (C.c_int, [C.c_void_p, _vpp]),
    "ptnn_swap_apply": (C.c_int, [C.c_void_p, _ip, C.c_int]),
    "ptnn_xchg_ptr": (C.c_int, [C.c_void_p, _vpp, _ip]),
    "ptnn_swap_pack": (C.c_int, [C.c_void_p, C.c_int]),
    "ptnn_swap_apply_gathered": (C.c_int, [C.c_void_p, C.c_int]),
    "ptnn_get_traces": (C.c_int, [C.c_void_p, C.c_int, C.c_int, _fp, _fp, _fp, _fp, _fp, _fp, _ip]),
    "ptnn_get_trace_rows": (C.c_int, [C.c_void_p, C.c_int, C.c_int, _fp]),
    "ptnn_get_swap_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), _ip]),
    "ptnn_get_swap_log": (C.c_int, [C.c_void_p, _ip, C.c_int]),
    "ptnn_get_state": (C.c_int, [C.c_void_p, _fp, _fp, _fp, _fp, _ip, _ip, _ip]),
    "ptnn_get_labels": (C.c_int, [C.c_void_p, _ip]),
    "ptnn_checkpoint_size": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "ptnn_checkpoint_save": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "ptnn_checkpoint_load": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "ptnn_evaluate": (C.c_int, [C.c_void_p, _fp, _fp, C.c_int, _fp]),
    "ptnn_langevin_gradient": (C.c_int, [C.c_void_p, _fp, C.c_int, _fp]),
    "ptnn_time_sgd_epoch": (C.c_int, [C.c_void_p, _fp, C.c_int, C.POINTER(C.c_double)]),
    "ptnn_time_tree_round": (C.c_int, [C.c_void_p, _fp, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    "ptnn_tape": (C.c_int, [C.c_void_p, C.c_int, C.c_int, _fp, _fp]),
    "ptnn_describe": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "ptnn_kernel_time": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    "ptnn_debug_stamps": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "ptnn_text_round": (C.c_int, [C.POINTER(C.c_double), C.c_int64, C.c_char_p]),
    "ptnn_savetxt": (C.c_int, [C.c_char_p, C.POINTER(C.c_double), C.c_int64, C.c_int64, C.c_char_p]),
    "ptnn_savetxt_f32": (C.c_int, [C.c_char_p, _fp, C.c_int64, C.c_int64, C.c_int64, C.c_char_p, C.c_int]),
    "ptnn_text_round_f32": (C.c_int, [_fp, C.POINTER(C.c_double), C.c_int64, C.c_char_p]),
    "ptnn_posterior_matrix": (C.c_int, [_fp, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.POINTER(C.c_double), C.c_int]),
    "ptnn_savetxt_f32_batch": (C.c_int, [C.c_int, C.POINTER(C.c_char_p), C.POINTER(_fp), C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                         C.POINTER(C.c_int64), C.POINTER(C.c_char_p), C.c_int, C.c_int]),
    "ptnn_trace_image": (C.c_int, [C.c_void_p, C.POINTER(_fp), _ip, C.POINTER(_fp)]),
    "ptnn_trace_image_fetch": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "ptnn_trace_image_wait": (C.c_int, [C.c_void_p, C.c_int]),
}


def load_library():
    """dlopen libptnn.so and declare every prototype.  Loading does not touch the GPU."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise PtnnError(f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                        f"(hipcc --offload-arch=gfx950); there is no CPU fallback")
    lib = C.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError here = header and library out of step
        fn.restype, fn.argtypes = res, args
    if lib.ptnn_abi_version() != ABI_VERSION:
        raise PtnnError(f"libptnn ABI {lib.ptnn_abi_version()} != binding {ABI_VERSION}")
    _lib = lib
    return lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a, typ=_fp):
    return None if a is None else a.ctypes.data_as(typ)


def comm_unique_id():
    """ncclGetUniqueId through libptnn (loads librccl.so): 128 bytes rank 0 hands to every rank."""
    lib = load_library()
    buf = C.create_string_buffer(UNIQUE_ID_BYTES)
    if lib.ptnn_comm_unique_id(buf, UNIQUE_ID_BYTES) < 0:
        raise PtnnError(lib.ptnn_last_error().decode())
    return buf.raw


def comm_probe(devices):
    """ptnn_comm_probe: one bounded RCCL round trip among `devices` IN THIS PROCESS -> seconds it took; raises PtnnError with the
    stage that failed.  Callers want distributed.rccl_probe(), which runs this in a fresh child process."""
    lib = load_library()
    dev = np.ascontiguousarray(devices, dtype=np.int32)
    sec = C.c_double()
    if lib.ptnn_comm_probe(_ptr(dev, _ip), int(dev.size), C.byref(sec)) < 0:
        raise PtnnError(lib.ptnn_last_error().decode())
    return sec.value


def comm_last_stage():
    """The last stage a communicator bring-up / exchange entered in this process (text), for error reports."""
    lib = load_library()
    buf = C.create_string_buffer(256)
    lib.ptnn_comm_last_stage(buf, 256)
    return buf.value.decode()


def route(src, n_local, rank):
    """ptnn_route: the rows `rank` receives / sends for the permutation src -> (recvs, sends), each a list of
    (local row, peer rank, global destination slot) in ascending global destination slot.  Pure host function."""
    lib = load_library()
    src = np.ascontiguousarray(src, dtype=np.int32)
    msg = np.empty((src.size + 1, 4), np.int32)
    n = lib.ptnn_route(_ptr(src, _ip), src.size, int(n_local), int(rank), _ptr(msg, _ip), msg.shape[0])
    if n < 0:
        raise PtnnError(lib.ptnn_last_error().decode())
    recvs = [(int(m[2]), int(m[1]), int(m[3])) for m in msg[:n] if m[0] == 0]
    sends = [(int(m[2]), int(m[1]), int(m[3])) for m in msg[:n] if m[0] == 1]
    return recvs, sends


class Sampler:
    """Thin object wrapper over a ptnn_handle (one GPU, one contiguous block of the ladder)."""

    def __init__(self, **kw):
        self.lib = load_library()
        cfg = Config()
        cfg.struct_bytes = C.sizeof(Config)
        for k, v in kw.items():
            setattr(cfg, k, v)
        self.cfg = cfg
        self.P = cfg.n_in * cfg.n_hidden + cfg.n_hidden * cfg.n_out + cfg.n_hidden + cfg.n_out
        self.R = cfg.n_replicas_local
        self.S = cfg.n_samples
        h = C.c_void_p()
        self._check(self.lib.ptnn_create(C.byref(cfg), C.byref(h)))
        self.h = h

    def _check(self, rc):
        if rc < 0:
            raise PtnnError(self.lib.ptnn_last_error().decode())
        return rc

    def close(self):
        if getattr(self, "h", None):
            self.lib.ptnn_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_data(self, train, test):
        tr, te = _f32(train), _f32(test)
        if tr.ndim != 2 or te.ndim != 2 or tr.shape[1] != te.shape[1]:
            raise ValueError("train/test must be 2-D with the same number of columns")
        self._check(self.lib.ptnn_set_data(self.h, _ptr(tr), tr.shape[0], _ptr(te), te.shape[0], tr.shape[1]))
        self.ntr, self.nte = tr.shape[0], te.shape[0]

    def set_state(self, w0, temperatures):
        w0, t = _f32(w0), _f32(temperatures)
        if w0.shape != (self.R, self.P) or t.shape != (self.R,):
            raise ValueError(f"w0 must be [{self.R},{self.P}], temperatures [{self.R}]")
        self._check(self.lib.ptnn_set_state(self.h, _ptr(w0), _ptr(t)))

    def set_ladder(self, temperatures_global):
        t = _f32(temperatures_global)
        if t.shape != (self.cfg.n_replicas_global,):
            raise ValueError("temperatures_global must have n_replicas_global entries")
        self._check(self.lib.ptnn_set_ladder(self.h, _ptr(t)))

    def run(self, n_steps=-1):
        self._check(self.lib.ptnn_run(self.h, int(n_steps)))

    def sync(self):
        self._check(self.lib.ptnn_sync(self.h))

    def steps_done(self):
        return self.lib.ptnn_steps_done(self.h)

    # ---- sharded ladder: communicators ----
    def comm_init(self, unique_id, rank, nranks):
        """RCCL communicator over the ranks that own the blocks of this ladder (collective: every rank calls it)."""
        self._check(self.lib.ptnn_comm_init(self.h, unique_id, len(unique_id), int(rank), int(nranks)))

    def comm_init_host(self, rank, nranks, all_gather, send_recv):
        """Host-staged transport: all_gather(buf: uint8 array [nranks, bytes_per_rank]) fills the other ranks' blocks in place;
        send_recv(msgs: list of (peer, is_send, uint8 array)) completes all messages.  Exceptions become an error return."""
        def _ag(ctx, buf, nbytes):
            try:
                arr = np.ctypeslib.as_array(C.cast(buf, C.POINTER(C.c_uint8)), shape=(int(nranks), int(nbytes)))
                all_gather(arr)
                return 0
            except Exception:                                   # noqa: BLE001 -- must not unwind through C
                import traceback
                traceback.print_exc()
                return -1

        def _sr(ctx, n, peer, is_send, bufs, nbytes):
            try:
                msgs = [(int(peer[k]), bool(is_send[k]),
                         np.ctypeslib.as_array(C.cast(bufs[k], C.POINTER(C.c_uint8)), shape=(int(nbytes),))) for k in range(n)]
                send_recv(msgs)
                return 0
            except Exception:                                   # noqa: BLE001
                import traceback
                traceback.print_exc()
                return -1
        self._callbacks = (ALL_GATHER_FN(_ag), SEND_RECV_FN(_sr))      # keep them alive as long as the handle
        self._check(self.lib.ptnn_comm_init_host(self.h, int(rank), int(nranks), C.cast(self._callbacks[0], C.c_void_p),
                                                 C.cast(self._callbacks[1], C.c_void_p), None))

    def comm_set_mode(self, mode):
        self._check(self.lib.ptnn_comm_set_mode(self.h, int(mode)))

    def comm_stats(self):
        a, b, r, m = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int32()
        self._check(self.lib.ptnn_comm_stats(self.h, C.byref(a), C.byref(b), C.byref(r), C.byref(m)))
        return dict(bytes_sent=a.value, bytes_received=b.value, rounds=r.value, mode={0: "none", 1: "gather", 2: "boundary"}[m.value])

    def comm_info(self):
        """What is attached to the handle: transport, rank, ranks as the communicator reports them, device (ptnn_comm_info)."""
        t, r, n, d = (C.c_int32() for _ in range(4))
        self._check(self.lib.ptnn_comm_info(self.h, C.byref(t), C.byref(r), C.byref(n), C.byref(d)))
        return dict(transport={0: "none", 1: "rccl", 2: "host"}[t.value], rank=r.value, nranks=n.value, device=d.value)

    def comm_finalize(self):
        self._check(self.lib.ptnn_comm_finalize(self.h))

    def run_segment(self):
        ho = C.c_int(0)
        self._check(self.lib.ptnn_run_segment(self.h, C.byref(ho)))
        return ho.value

    def swap_L_ptr(self, phantom):
        p = C.c_void_p()
        self._check(self.lib.ptnn_swap_L_ptr(self.h, int(phantom), C.byref(p)))
        return p.value

    def swap_set_L(self, L, phantom=0):
        L = _f32(L)
        if L.shape != (self.cfg.n_replicas_global,):
            raise ValueError("L must have n_replicas_global entries")
        self._check(self.lib.ptnn_swap_set_L(self.h, int(phantom), _ptr(L)))

    def swap_cascade(self, phantom):
        src = np.empty(self.cfg.n_replicas_global, dtype=np.int32)
        self._check(self.lib.ptnn_swap_cascade(self.h, int(phantom), _ptr(src, _ip)))
        return src

    def swap_row_ptr(self, local_replica):
        a, b = C.c_void_p(), C.c_void_p()
        self._check(self.lib.ptnn_swap_row_ptr(self.h, int(local_replica), C.byref(a), C.byref(b)))
        return a.value, b.value

    def stream_ptr(self):
        p = C.c_void_p()
        self._check(self.lib.ptnn_stream(self.h, C.byref(p)))
        return p.value

    def state_row_floats(self):
        return self.lib.ptnn_state_row_floats(self.h)

    def swap_apply(self, src, phantom):
        src = np.ascontiguousarray(src, dtype=np.int32)
        self._check(self.lib.ptnn_swap_apply(self.h, _ptr(src, _ip), int(phantom)))

    def xchg_ptr(self):
        base, n = C.c_void_p(), C.c_int32()
        self._check(self.lib.ptnn_xchg_ptr(self.h, C.byref(base), C.byref(n)))
        return base.value, n.value

    def swap_pack(self, phantom):
        self._check(self.lib.ptnn_swap_pack(self.h, int(phantom)))

    def swap_apply_gathered(self, phantom):
        self._check(self.lib.ptnn_swap_apply_gathered(self.h, int(phantom)))

    def traces(self, step0=0, nsteps=None, pos_w=True):
        n = self.S - step0 if nsteps is None else nsteps
        out = {
            "pos_w": np.empty((self.R, n, self.P), np.float32) if pos_w else None,
            "likeh": np.empty((self.R, n), np.float32),
            "rmse_train": np.empty((self.R, n), np.float32), "rmse_test": np.empty((self.R, n), np.float32),
            "acc_train": np.empty((self.R, n), np.float32), "acc_test": np.empty((self.R, n), np.float32),
            "accept": np.empty((self.R, n), np.int32),
        }
        self._check(self.lib.ptnn_get_traces(self.h, step0, n, _ptr(out["pos_w"]), _ptr(out["likeh"]),
                                             _ptr(out["rmse_train"]), _ptr(out["rmse_test"]), _ptr(out["acc_train"]),
                                             _ptr(out["acc_test"]), _ptr(out["accept"], _ip)))
        return out

    # ---- trace images: the download overlapped with sampling (ptnn_trace_image*) ----
    def trace_image(self):
        """(pos_w, rows): numpy views of the handle's pinned host images -- pos_w [R, S, P] float32 (rows strided by the device's
        padded row), rows [R, S, 8] as trace_rows() describes them.  Valid until close(); filled by trace_fetch()."""
        pw, rw, rf = _fp(), _fp(), C.c_int32()
        self._check(self.lib.ptnn_trace_image(self.h, C.byref(pw), C.byref(rf), C.byref(rw)))
        pos = np.ctypeslib.as_array(pw, shape=(self.R, self.S, rf.value))[:, :, :self.P]
        rows = np.ctypeslib.as_array(rw, shape=(self.R, self.S, 8))
        return pos, rows

    def trace_fetch(self, row0, nrows):
        """Queues the copy of trace rows [row0, row0 + nrows) into the images behind the steps queued so far; returns a ticket."""
        return self._check(self.lib.ptnn_trace_image_fetch(self.h, int(row0), int(nrows)))

    def trace_wait(self, ticket):
        self._check(self.lib.ptnn_trace_image_wait(self.h, int(ticket)))

    def trace_rows(self, row0=0, nrows=None):
        """The scalar trace rows as the device keeps them (ptnn_get_trace_rows), [R, nrows, 8] float32."""
        n = self.S - row0 if nrows is None else nrows
        rows = np.empty((self.R, n, 8), np.float32)
        self._check(self.lib.ptnn_get_trace_rows(self.h, int(row0), int(n), _ptr(rows)))
        return rows

    def eta_trace(self):
        """Regression: eta = log tau^2 of the state recorded in every trace row, [R, S] (row i + 1 after MH step i; 0 before the
        first accepted step); classification has no eta: None."""
        if self.cfg.task != TASK_REG:
            return None
        return self.trace_rows()[:, :, 3].copy()

    def log_alpha(self, step0=0, nsteps=None):
        """log alpha of MH steps step0 .. step0+nsteps-1 as the kernel computed it, [R, nsteps] (row i + 1 belongs to step i)."""
        n = self.S - 1 - step0 if nsteps is None else nsteps
        rows = np.empty((self.R, n, 8), np.float32)
        self._check(self.lib.ptnn_get_trace_rows(self.h, step0 + 1, n, _ptr(rows)))
        return rows[:, :, 6].copy()

    def swap_stats(self):
        a, b, r = C.c_int64(), C.c_int64(), C.c_int32()
        self._check(self.lib.ptnn_get_swap_stats(self.h, C.byref(a), C.byref(b), C.byref(r)))
        return a.value, b.value, r.value

    def swap_log(self, max_rounds=None):
        Rg = self.cfg.n_replicas_global
        cap = max_rounds or (self.S // self.cfg.swap_interval + 2)
        buf = np.empty((cap, Rg), np.int32)
        n = self._check(self.lib.ptnn_get_swap_log(self.h, _ptr(buf, _ip), cap))
        return buf[:n]

    def labels(self):
        lab = np.empty(self.cfg.n_replicas_global, np.int32)
        self._check(self.lib.ptnn_get_labels(self.h, _ptr(lab, _ip)))
        return lab

    def state(self):
        w = np.empty((self.R, self.P), np.float32)
        eta, lik, pri = (np.empty(self.R, np.float32) for _ in range(3))
        nacc, lg, lga = (np.zeros(self.R, np.int32) for _ in range(3))
        self._check(self.lib.ptnn_get_state(self.h, _ptr(w), _ptr(eta), _ptr(lik), _ptr(pri), _ptr(nacc, _ip), _ptr(lg, _ip),
                                            _ptr(lga, _ip)))
        return dict(w=w, eta=eta, likelihood=lik, prior=pri, num_accepted=nacc, langevin_count=lg, langevin_accepted=lga)

    def checkpoint(self):
        """State of the chains as bytes (ptnn_checkpoint_save); traces are not included."""
        n = C.c_int64()
        self._check(self.lib.ptnn_checkpoint_size(self.h, C.byref(n)))
        buf = np.empty(n.value, np.uint8)
        self._check(self.lib.ptnn_checkpoint_save(self.h, buf.ctypes.data_as(C.c_void_p), n.value))
        return buf.tobytes()

    def restore(self, blob):
        """Continue chains saved by checkpoint() (call after set_data, instead of set_state)."""
        buf = np.frombuffer(blob, np.uint8)
        self._check(self.lib.ptnn_checkpoint_load(self.h, buf.ctypes.data_as(C.c_void_p), buf.size))

    def evaluate(self, w, tau_sq=None):
        w = _f32(np.atleast_2d(w))
        n = w.shape[0]
        tau = None if tau_sq is None else _f32(np.broadcast_to(np.asarray(tau_sq, dtype=np.float32), (n,)))
        out = np.empty((n, 8), np.float32)
        self._check(self.lib.ptnn_evaluate(self.h, _ptr(w), _ptr(tau), n, _ptr(out)))
        return out

    def langevin_gradient(self, w):
        w = _f32(np.atleast_2d(w))
        out = np.empty_like(w)
        self._check(self.lib.ptnn_langevin_gradient(self.h, _ptr(w), w.shape[0], _ptr(out)))
        return out

    def time_sgd_epoch(self, w, reps=200, pair=False):
        """Milliseconds one sequential SGD epoch of one chain takes on the device (in-kernel constant-rate counter); pair=True
        (wide nets): (one epoch, a pair of epochs through one row loop)."""
        w = _f32(w).reshape(-1)
        ms = (C.c_double * 2)()
        self._check(self.lib.ptnn_time_sgd_epoch(self.h, _ptr(w), int(reps), ms))
        return (ms[0], ms[1]) if pair else ms[0]

    def time_tree_round(self, w, reps=200, xcd_local=True):
        """(forward pass ms, one granule one way ms, went through the XCD's L2?) -- ptnn_time_tree_round."""
        w = _f32(w).reshape(-1)
        ms = (C.c_double * 3)()
        self._check(self.lib.ptnn_time_tree_round(self.h, _ptr(w), int(reps), int(bool(xcd_local)), ms))
        return ms[0], ms[1], bool(ms[2])

    def tape(self, replica, step):
        noise, scal = np.empty(self.P, np.float32), np.empty(3, np.float32)
        self._check(self.lib.ptnn_tape(self.h, int(replica), int(step), _ptr(noise), _ptr(scal)))
        return noise, scal

    def debug_stamps(self):
        buf = (C.c_uint64 * 160)()
        self._check(self.lib.ptnn_debug_stamps(self.h, buf))
        return list(buf)

    def describe(self):
        """What the handle launches (kernel, grid, LDS, occupancy) as a dict; see ptnn_describe."""
        import json
        buf = C.create_string_buffer(2048)
        self._check(self.lib.ptnn_describe(self.h, buf, len(buf)))
        return json.loads(buf.value.decode())

    def kernel_time(self, reset=False):
        n, ms = C.c_int64(), C.c_double()
        self._check(self.lib.ptnn_kernel_time(self.h, int(reset), C.byref(n), C.byref(ms)))
        return n.value, ms.value


def savetxt(path, array, fmt, append=False):
    """np.savetxt(path, array, fmt=fmt) for 1-D / 2-D arrays, byte for byte, formatted by the C library (GIL released).  float32
    arrays (the device's traces, also row-strided views of them) are written as they are; anything else goes through float64."""
    lib = load_library()
    a = np.asarray(array)
    if a.ndim not in (1, 2):
        raise ValueError("savetxt handles 1-D and 2-D arrays")
    if a.dtype == np.float32 and a.size and a.strides[-1] == 4 and (a.ndim == 1 or (a.strides[0] % 4 == 0 and a.strides[0] >= 4 * a.shape[1])):
        rows, cols, stride = (a.shape[0], 1, 1) if a.ndim == 1 else (a.shape[0], a.shape[1], a.strides[0] // 4)
        rc = lib.ptnn_savetxt_f32(os.fsencode(path), a.ctypes.data_as(_fp), rows, cols, stride, fmt.encode(), int(bool(append)))
    else:
        if append:
            raise ValueError("append mode takes float32 data")
        a = np.ascontiguousarray(a, dtype=np.float64)
        rows, cols = (a.shape[0], 1) if a.ndim == 1 else a.shape
        rc = lib.ptnn_savetxt(os.fsencode(path), a.ctypes.data_as(C.POINTER(C.c_double)), rows, cols, fmt.encode())
    if rc < 0:
        raise PtnnError(lib.ptnn_last_error().decode())


def savetxt_batch(jobs, append=False, threads=8):
    """savetxt(path, array, fmt) for every (path, array, fmt) of `jobs` in ONE call into the C library, which spreads the files over
    `threads` host threads in the order given.  float32 arrays (1-D, or 2-D with unit column stride) only: the per-chain files of a
    window of trace rows."""
    lib = load_library()
    n = len(jobs)
    if n == 0:
        return
    paths, data, fmts = (C.c_char_p * n)(), (_fp * n)(), (C.c_char_p * n)()
    rows, cols, strides = (C.c_int64 * n)(), (C.c_int64 * n)(), (C.c_int64 * n)()
    keep = []
    for k, (path, array, fmt) in enumerate(jobs):
        a = np.asarray(array)
        if a.dtype != np.float32 or a.ndim not in (1, 2) or a.strides[-1] != 4 or (a.ndim == 2 and (a.strides[0] % 4 or a.strides[0] < 4 * a.shape[1])):
            a = np.ascontiguousarray(a, dtype=np.float32)
            if a.ndim not in (1, 2):
                raise ValueError("savetxt handles 1-D and 2-D arrays")
        keep.append(a)
        paths[k], fmts[k] = os.fsencode(path), fmt.encode()
        data[k] = a.ctypes.data_as(_fp)
        rows[k], cols[k], strides[k] = (a.shape[0], 1, 1) if a.ndim == 1 else (a.shape[0], a.shape[1], a.strides[0] // 4)
    if lib.ptnn_savetxt_f32_batch(n, paths, data, rows, cols, strides, fmts, int(bool(append)), int(threads)) < 0:
        raise PtnnError(lib.ptnn_last_error().decode())


def _chunks(n, threads, grain=65536):
    k = max(1, min(int(threads), n // grain + 1))
    b = np.linspace(0, n, k + 1).astype(np.int64)
    return [(int(b[i]), int(b[i + 1])) for i in range(k) if b[i + 1] > b[i]]


def text_round(array, fmt, threads=8):
    """Values as np.loadtxt reads them back after np.savetxt(..., fmt=fmt), float64, in the C library (exact integer arithmetic
    for the '%1.Nf' formats, no strings), chunked over a few threads (ctypes releases the GIL)."""
    from concurrent.futures import ThreadPoolExecutor
    lib = load_library()
    src = np.asarray(array)
    f32 = src.dtype == np.float32
    src = np.ascontiguousarray(src, dtype=np.float32 if f32 else np.float64)
    out = np.empty(src.shape, dtype=np.float64) if f32 else np.array(src, copy=True)
    n = out.size
    if n == 0:
        return out
    flat_in, flat = src.reshape(-1), out.reshape(-1)
    dp = C.POINTER(C.c_double)

    def work(rng):
        lo, hi = rng
        if f32:
            rc = lib.ptnn_text_round_f32(flat_in[lo:hi].ctypes.data_as(_fp), flat[lo:hi].ctypes.data_as(dp), hi - lo, fmt.encode())
        else:
            rc = lib.ptnn_text_round(flat[lo:hi].ctypes.data_as(dp), hi - lo, fmt.encode())
        if rc < 0:
            raise PtnnError(lib.ptnn_last_error().decode())
    parts = _chunks(n, threads)
    if len(parts) == 1:
        work(parts[0])
    else:
        with ThreadPoolExecutor(max_workers=len(parts)) as ex:
            list(ex.map(work, parts))
    return out


def posterior_matrix(pos_w, first_row, threads=8):
    """pos_w float32 [R, S, P] -> float64 [P, R * (S - first_row)]: rows from first_row on, chains side by side, transposed
    (what show_results returns as pos_w, REG:795-797, 848)."""
    lib = load_library()
    a = np.asarray(pos_w)
    R, S, P = a.shape
    # the padded rows of a trace image are read in place; anything else is made dense first
    if not (a.dtype == np.float32 and a.strides[2] == 4 and a.strides[1] % 4 == 0 and a.strides[1] >= 4 * P and a.strides[0] == S * a.strides[1]):
        a = np.ascontiguousarray(a, dtype=np.float32)
    out = np.empty((P, R * (S - first_row)), dtype=np.float64)
    if lib.ptnn_posterior_matrix(a.ctypes.data_as(_fp), R, S, P, a.strides[1] // 4, int(first_row), out.ctypes.data_as(C.POINTER(C.c_double)),
                                 int(threads)) < 0:
        raise PtnnError(lib.ptnn_last_error().decode())
    return out
