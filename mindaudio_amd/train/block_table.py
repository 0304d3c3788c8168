"""Host side of the block launch table (include/mindaudio_amd.h, "a Conformer block's launches from one C call").

`BlockTable.recording()` makes `_lib.load()` hand out a proxy for the duration of one walked step: every replayable C-ABI call made
between `segment(backward, block)` marks is issued as usual AND appended to the table, and every tensor the kernel wrappers allocate
meanwhile is kept referenced by the table (so that the caching allocator can never hand a recorded address to anybody else).  From the
next step of that batch shape on the engine issues each block with one `ma_conformer_block_fwd_train` / `_bwd_train` call.

What is recorded is the reference's training step over the encoder blocks (/root/reference/mindaudio/utils/train_one_step.py:13-48,
models/conformer.py:109-156); the recorder adds no arithmetic of its own."""
import ctypes
import struct

from .. import _lib


class _KeepTorch:
    """`torch`, with the allocating calls of the kernel wrappers remembered (train/kernels.py allocates through `_t()`)."""

    def __init__(self, torch, keep):
        self._torch, self._keep = torch, keep

    def __getattr__(self, name):
        return getattr(self._torch, name)

    def _held(self, x):
        self._keep.append(x)
        return x

    def empty(self, *a, **k):
        return self._held(self._torch.empty(*a, **k))

    def zeros(self, *a, **k):
        return self._held(self._torch.zeros(*a, **k))

    def empty_like(self, *a, **k):
        return self._held(self._torch.empty_like(*a, **k))

    def zeros_like(self, *a, **k):
        return self._held(self._torch.zeros_like(*a, **k))


class _RecordingLib:
    """The library, with replayable calls copied into the table's current segment."""

    def __init__(self, lib, table):
        self._lib, self._table, self._wrapped = lib, table, {}

    def __getattr__(self, name):
        fn = self._wrapped.get(name)
        if fn is None:
            fn = self._wrapped[name] = self._table._wrap(name, getattr(self._lib, name))
        return fn


def _as_word(argtype, arg):
    if argtype is ctypes.c_void_p:
        if arg is None:
            return 0
        return int(arg.value or 0) if isinstance(arg, ctypes.c_void_p) else int(arg)
    if argtype in (ctypes.c_float, ctypes.c_double):
        return struct.unpack("<q", struct.pack("<d", float(arg)))[0]
    v = int(arg.value) if hasattr(arg, "value") else int(arg)
    return v - (1 << 64) if v >= (1 << 63) else v


class BlockTable:
    def __init__(self):
        self.lib = _lib.load()
        self.handle = ctypes.c_void_p(self.lib.ma_block_table_create())
        if not self.handle:
            raise _lib.MindaudioAmdError("ma_block_table_create failed")
        self.keep = []          # tensors named by the entries (and everything allocated while recording)
        self.where = None       # (backward, block) being recorded
        self.seed = None        # the recorded step's dropout seed: every seed argument must be this (or 0: no dropout)
        self.recorded = 0
        self.broken = None      # why this recording cannot be replayed (the walked step it was taken from is complete all the same)

    def __del__(self):
        try:
            if self.handle:
                self.lib.ma_block_table_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def clear(self):
        """Give up every tensor this table keeps alive (the recorded entries then name freed memory: the table must not be replayed
        again - the engine drops it in the same breath)."""
        del self.keep[:]  # (in place: the allocation proxy holds the same list)
        self.recorded, self.broken = 0, self.broken or "cleared"

    # ---- recording ---------------------------------------------------------------------------------------------------------------
    class _Recording:
        def __init__(self, table, seed):
            self.table, self.seed = table, seed

        def __enter__(self):
            t = self.table
            if _lib.recording() is not None:
                raise _lib.MindaudioAmdError("a block table is already being recorded by this thread")
            t.seed, t.where = int(self.seed), None
            t._proxy = _RecordingLib(t.lib, t)
            _lib.set_recording(t)
            return t

        def __exit__(self, *exc):
            _lib.set_recording(None)
            self.table.where = None
            return False

    def recording(self, seed):
        return BlockTable._Recording(self, seed)

    def segment(self, backward, block):
        """The calls that follow belong to (direction, block); None: not to any block (they are issued but not recorded)."""
        self.where = None if backward is None else (1 if backward else 0, int(block))

    def torch(self, torch):
        kt = self.__dict__.get("_keep_torch")
        if kt is None:
            kt = self._keep_torch = _KeepTorch(torch, self.keep)
        return kt

    def _wrap(self, name, fn):
        proto = _lib.PROTOTYPES.get(name)
        fid = int(self.lib.ma_block_table_entry_point(name.encode())) if proto is not None else -1
        if fid < 0:
            if fid != -2 or name.startswith("ma_block_table_") or name.startswith("ma_conformer_block_"):
                return fn  # size queries, layout helpers, the table's own entry points: nothing to replay

            def refuse(*args):  # issued as usual; the table is marked: the engine keeps walking this batch shape
                if self.where is not None and self.broken is None:
                    self.broken = "%s cannot be replayed from a block table" % name
                return fn(*args)
            return refuse
        argtypes = proto[1]
        seed_mask = int(self.lib.ma_block_table_entry_point_seeds(fid))

        def call(*args):
            rc = fn(*args)
            if self.where is not None and rc == 0 and self.broken is None:
                try:
                    self._add(name, fid, argtypes, seed_mask, args)
                except _lib.MindaudioAmdError as e:  # (the launch itself is done: only the RECORDING failed)
                    self.broken = str(e)
            return rc
        return call

    def _add(self, name, fid, argtypes, seed_mask, args):
        if len(args) != len(argtypes):
            raise _lib.MindaudioAmdError("%s: %d arguments for %d parameters" % (name, len(args), len(argtypes)))
        words, blob = [], bytearray()
        for k, (tp, a) in enumerate(zip(argtypes, args)):
            if isinstance(tp, type) and issubclass(tp, ctypes._Pointer):  # a host struct (or host array of structs)
                if a is None:
                    words.append(-1)
                    continue
                obj = getattr(a, "_obj", None)  # byref(x)
                if obj is None:
                    obj = a.contents if isinstance(a, ctypes._Pointer) else a
                if isinstance(obj, _lib.TrainEpilogue) and obj.seed not in (0, self.seed):
                    raise _lib.MindaudioAmdError("%s: an epilogue with seed %d in a step of seed %d" % (name, obj.seed, self.seed))
                words.append(len(blob))
                raw = bytes(obj)
                blob += raw + b"\0" * (-len(raw) % 8)
                continue
            w = _as_word(tp, a)
            if (seed_mask >> k) & 1 and w not in (0, self.seed):
                raise _lib.MindaudioAmdError("%s: seed argument %d in a step of seed %d" % (name, w, self.seed))
            words.append(w)
        arr = (ctypes.c_int64 * len(words))(*words)
        buf = (ctypes.c_char * len(blob)).from_buffer(blob) if blob else None
        _lib.check(self.lib.ma_block_table_add(self.handle, self.where[0], self.where[1], fid, arr, len(words), buf, len(blob)),
                   "block_table_add(%s)" % name)
        self.recorded += 1

    def nbytes(self):
        """Device bytes this table keeps alive (distinct storages of everything allocated while it was recorded)."""
        seen, total = set(), 0
        for x in self.keep:
            try:
                st = x.untyped_storage()
                key, n = st.data_ptr(), st.nbytes()
            except Exception:
                continue
            if key not in seen:
                seen.add(key)
                total += n
        return total

    # ---- replay ------------------------------------------------------------------------------------------------------------------
    def calls(self, backward, block):
        return int(self.lib.ma_block_table_calls(self.handle, 1 if backward else 0, block))

    def forward(self, block, seed, stream):
        rc = self.lib.ma_conformer_block_fwd_train(self.handle, block, seed, stream)
        if rc != 0:
            _lib.check(rc, "conformer_block_fwd_train(block %d, call %d)" % (block, self.lib.ma_block_table_failed_call(self.handle)))

    def backward(self, block, seed, stream):
        rc = self.lib.ma_conformer_block_bwd_train(self.handle, block, seed, stream)
        if rc != 0:
            _lib.check(rc, "conformer_block_bwd_train(block %d, call %d)" % (block, self.lib.ma_block_table_failed_call(self.handle)))
