"""Chunk-loader pieces on the native reader: the functions keep the names and the semantics of the
reference's ``data_io`` so that a loader written against it reads the same.

    read_mat_ark / read_mat_scp   data_io.py:1039-1085  (binary float / double / compressed matrices, plain files)
    read_vec_int_ark              data_io.py:790-838    (alignment / pdf-id vectors)
    context_window                data_io.py:228-241
    normalize_chunk               data_io.py:263        (mean / variance normalisation of the concatenated chunk)
    finish_chunk                  data_io.py:244-274    (load_chunk after load_dataset: splice, normalise, label shift,
                                                         column-stack; returns float32 where the reference carries
                                                         float64 until run_nn's .float())

The Kaldi pipelines inside ``fea_opts`` / ``lab_opts`` (apply-cmvn, add-deltas, ali-to-pdf ...) are external programs and
stay with the reference's reader; these functions cover tables that already exist on disk.
"""
import ctypes

import numpy as np
import torch

from . import _lib

_KEYCAP = 4096


def _fp(a):
    return ctypes.c_void_p(a.ctypes.data)


class _Table:
    def __init__(self, path, offset=0):
        self.lib = _lib.load()
        self.h = self.lib.pk_ark_open(str(path).encode(), int(offset))
        if not self.h:
            raise IOError(self.lib.pk_last_error().decode())
        self.key = ctypes.create_string_buffer(_KEYCAP)

    def next(self, key_expected=True):
        rows, cols = ctypes.c_int64(), ctypes.c_int64()
        rc = self.lib.pk_ark_next(self.h, int(key_expected), self.key, _KEYCAP, ctypes.byref(rows), ctypes.byref(cols))
        if rc == 0:
            return None
        if rc != 1:
            raise IOError(self.lib.pk_last_error().decode())
        mat = np.empty((rows.value, cols.value), dtype=np.float32)
        _lib.check(self.lib.pk_ark_read(self.h, _fp(mat)), "pk_ark_read")
        return self.key.value.decode("latin1"), mat

    def close(self):
        if self.h:
            self.lib.pk_ark_close(self.h)
            self.h = None

    def __del__(self):
        self.close()


def read_mat_ark(path):
    """generator(key, mat) over a binary ark file ("ark:" prefix allowed), data_io.py:1062-1085."""
    path = str(path)
    if path.startswith("ark:"):
        path = path[4:]
    t = _Table(path)
    try:
        while True:
            item = t.next(True)
            if item is None:
                return
            yield item
    finally:
        t.close()


def read_vec_int_ark(path):
    """generator(key, int32 vector) over a binary integer-vector ark (alignments / pdf ids), data_io.py:790-808."""
    path = str(path)
    if path.startswith("ark:"):
        path = path[4:]
    t = _Table(path)
    try:
        while True:
            n = ctypes.c_int64()
            rc = t.lib.pk_ivec_next(t.h, t.key, _KEYCAP, ctypes.byref(n))
            if rc == 0:
                return
            if rc != 1:
                raise IOError(t.lib.pk_last_error().decode())
            vec = np.empty(n.value, dtype=np.int32)
            _lib.check(t.lib.pk_ivec_read(t.h, _fp(vec)), "pk_ivec_read")
            yield t.key.value.decode("latin1"), vec
    finally:
        t.close()


read_ali_ark = read_vec_int_ark  # data_io.py:785-787


def read_mat(rxfile):
    """One matrix from "file" or "file:offset" (an scp entry), data_io.py:1087-1104."""
    rxfile = str(rxfile).strip()
    if rxfile.startswith("ark:"):
        rxfile = rxfile[4:]
    offset = 0
    head, sep, tail = rxfile.rpartition(":")
    if sep and tail.isdigit():
        rxfile, offset = head, int(tail)
    t = _Table(rxfile, offset)
    try:
        item = t.next(False)
        if item is None:
            raise IOError("read_mat: no matrix at %s:%d" % (rxfile, offset))
        return item[1]
    finally:
        t.close()


def read_mat_scp(path):
    """generator(key, mat) over a Kaldi scp ("key file[:offset]" lines), data_io.py:1039-1060."""
    with open(path, "r") as f:
        for line in f:
            line = line.strip()
            if not line:
                continue
            key, rxfile = line.split(" ", 1)
            yield key, read_mat(rxfile)


def context_window(fea, left, right):
    """data_io.py:228-241 on a (rows, cols) array -> (rows-left-right, cols*(left+right+1)) float32."""
    fea = np.ascontiguousarray(fea, dtype=np.float32)
    rows, cols = fea.shape
    if rows < left + right:
        raise ValueError("context_window: %d rows cannot hold a -%d..+%d window" % (rows, left, right))
    out = np.empty((rows - left - right, cols * (left + right + 1)), dtype=np.float32)
    _lib.check(_lib.load().pk_context_window(_fp(fea), rows, cols, int(left), int(right), _fp(out)), "pk_context_window")
    return out


def normalize_chunk(data_set):
    """(x - mean) / std per column over the whole chunk, in place on a float32 C-contiguous array (data_io.py:263)."""
    if data_set.dtype != np.float32 or not data_set.flags["C_CONTIGUOUS"]:
        raise ValueError("normalize_chunk works in place on a C-contiguous float32 array")
    _lib.check(_lib.load().pk_mean_var_norm(_fp(data_set), data_set.shape[0], data_set.shape[1]), "pk_mean_var_norm")
    return data_set


def finish_chunk(data_set, data_lab, end_index_fea, left, right):
    """What load_chunk does with load_dataset's output (data_io.py:253-274): context window, end-index shift, chunk
    normalisation, label shift / trim, column_stack.  Returns (data_set float32 [rows, feat+1], end_index_fea)."""
    data_set = np.ascontiguousarray(data_set, dtype=np.float32)
    if left != 0 or right != 0:
        data_set = context_window(data_set, left, right)
    end_index_fea = np.asarray(end_index_fea).copy() - left
    end_index_fea[-1] = end_index_fea[-1] - right
    data_set = normalize_chunk(data_set if data_set.flags["WRITEABLE"] else data_set.copy())
    data_lab = np.asarray(data_lab)
    data_lab = data_lab - data_lab.min()
    data_lab = data_lab[left:-right] if right > 0 else data_lab[left:]
    return np.column_stack((data_set, data_lab.astype(np.float32))), end_index_fea


def finish_chunk_device(data_set, data_lab, end_index_fea, left, right, device):
    """finish_chunk with the splicing and the normalisation done on `device`: the UN-spliced features cross PCIe
    (1/(left+right+1) of the bytes the reference uploads: 176 MB instead of 1.9 GB for a 1.1 M-frame fMLLR chunk with
    an 11-frame window) and the chunk is born resident, which is what run_nn_dp's BatchAssembler gathers from.
    Statistics are accumulated in float64 like the reference's (data_io.py:263).  Returns (tensor [rows, feat+1], end_index)."""
    x = torch.as_tensor(np.ascontiguousarray(data_set, dtype=np.float32)).to(device, non_blocking=True)
    rows, W = x.shape[0], left + right + 1
    if rows < left + right:
        raise ValueError("finish_chunk_device: %d rows cannot hold a -%d..+%d window" % (rows, left, right))
    if W > 1:
        x = torch.cat([x[b:rows - left - right + b] for b in range(W)], 1)
    xd = x.double()
    mean = xd.mean(0)
    std = (xd - mean).pow(2).mean(0).sqrt()
    x = ((xd - mean) / std).float()
    end_index_fea = np.asarray(end_index_fea).copy() - left
    end_index_fea[-1] = end_index_fea[-1] - right
    lab = np.asarray(data_lab)
    lab = lab - lab.min()
    lab = lab[left:-right] if right > 0 else lab[left:]
    lab_t = torch.as_tensor(lab.astype(np.float32)).to(device, non_blocking=True)
    return torch.cat((x, lab_t[:, None]), 1), end_index_fea
