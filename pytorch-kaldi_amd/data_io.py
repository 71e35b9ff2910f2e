"""Chunk-loader pieces on the native reader: the functions keep the names and the semantics of the
reference's ``data_io`` so that a loader written against it reads the same.

    read_mat_ark / read_mat_scp   data_io.py:1039-1085  (binary float / double / compressed matrices, plain files)
    read_vec_int_ark              data_io.py:790-838    (alignment / pdf-id vectors)
    context_window                data_io.py:228-241
    normalize_chunk               data_io.py:263        (mean / variance normalisation of the concatenated chunk)
    finish_chunk                  data_io.py:244-274    (load_chunk after load_dataset: splice, normalise, label shift,
                                                         column-stack; returns float32 where the reference carries
                                                         float64 until run_nn's .float())
    load_dataset                  data_io.py:57-225     (everything load_dataset does AFTER its two Kaldi pipes: key
                                                         filtering, splitting of long sentences, the two length sorts,
                                                         concatenation, end indices)
    load_chunk                    data_io.py:244-274    (tables on disk -> [names, data_set, end_index]; takes the
                                                         feature scp / ark and the alignment ark(s) directly instead of
                                                         the fea_opts / lab_opts command strings)
    read_lab_fea                  data_io.py:536-655    (the chunk reader core.run_nn starts in its reader thread, same
                                                         signature and shared_list contract, for chunk cfg files whose
                                                         fea_opts are empty: several feature streams with different
                                                         context windows, several label sets, shuffle)
    dict_fea_lab_arch             utils.py:1889-1994    (which streams / labels / architectures the [model] lines use)

The Kaldi pipelines inside ``fea_opts`` / ``lab_opts`` (apply-cmvn, add-deltas, ali-to-pdf ...) are external programs and
stay with the reference's reader; these functions cover tables that already exist on disk.
"""
import ctypes
import glob
import gzip
import configparser
import os
import re
import shutil
import tempfile

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
    """generator(key, int32 vector) over a binary integer-vector ark (alignments / pdf ids), data_io.py:790-808.
    A ".gz" file is inflated to a temporary file first (the reference pipes `gunzip -c`, data_io.py:45-47)."""
    path = str(path)
    if path.startswith("ark:"):
        path = path[4:]
    tmp = None
    if path.endswith(".gz"):
        with gzip.open(path, "rb") as z, tempfile.NamedTemporaryFile(prefix="pk_ali_", suffix=".ark", delete=False) as f:
            shutil.copyfileobj(z, f)
            tmp = path = f.name
    try:
        t = _Table(path)
    except Exception:
        if tmp:
            os.unlink(tmp)
        raise
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
        if tmp:
            os.unlink(tmp)


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


def _chunk_config(max_sequence_length):
    """data_io.py:114-127: an int means the same size and step for features and labels."""
    if isinstance(max_sequence_length, dict):
        return tuple(int(max_sequence_length[k]) for k in ("chunk_size_fea", "chunk_step_fea", "chunk_size_lab", "chunk_step_lab"))
    if isinstance(max_sequence_length, (int, np.integer)):
        m = int(max_sequence_length)
        return m, m, m, m
    raise ValueError("Unknown type of max_sequence_length")


def _split_sentence(fea, lab, cfg, fea_only):
    """data_io.py:70-112: a sentence longer than the chunk size is cut into chunks of that size; the tail stays with
    the last chunk unless it is longer than a quarter of the size (500 -> a 625-frame sentence is one 500 + one 125)."""
    size_f, step_f, size_l, step_l = cfg
    zeros = lambda f: np.zeros((f.shape[0],))  # noqa: E731
    if not (len(fea) > size_f and size_f > 0):
        return [fea], [zeros(fea) if fea_only else lab]
    feas, labs = [], []
    threshold = size_f + size_f / 4
    for i in range((len(fea) + size_f - 1) // size_f):
        start = i * step_f
        last = not (len(fea) - start > threshold) if start < len(fea) else True
        f = fea[start:] if last else fea[start:start + size_f]
        if fea_only:
            l = zeros(f)
        else:
            l = lab[i * step_l:] if last else lab[i * step_l:i * step_l + size_l]
        feas.append(f)
        labs.append(l)
        if last:
            break
    return feas, labs


def load_dataset(fea, lab, max_sequence_length, fea_only=False):
    """What the reference's load_dataset does once its Kaldi pipes have delivered (data_io.py:41-54, 57-225).
    fea: {key: (frames, dim) array}; lab: {key: integer vector} (ignored when fea_only).  Returns the reference's
    [names, fea_conc, lab_conc, end_index_fea, end_index_lab]:
      * alignments without features and features without alignments are dropped (:41-54);
      * sentences are visited by (length, key), long ones split (:129-141), the chunks then stably re-sorted by
        length and concatenated (:143-164).  As in the reference, `names` keeps the order BEFORE that second sort."""
    if not fea_only:
        lab = {k: v for k, v in lab.items() if k in fea}
        fea = {k: v for k, v in fea.items() if k in lab}
    cfg = _chunk_config(max_sequence_length)
    names, feas, labs = [], [], []
    for k in sorted(sorted(fea.keys()), key=lambda k: len(fea[k])):
        f_chunks, l_chunks = _split_sentence(fea[k], None if fea_only else lab[k], cfg, fea_only)
        for j in range(len(f_chunks)):
            feas.append(f_chunks[j])
            labs.append(l_chunks[j])
            names.append(k + "_split" + str(j) if len(f_chunks) > 1 else k)
    if not feas:
        raise ValueError("load_dataset: no sentence has both features and an alignment")
    order = sorted(range(len(feas)), key=lambda i: feas[i].shape[0])  # stable, like sorted() on the zipped pairs
    feas, labs = [feas[i] for i in order], [labs[i] for i in order]
    end_fea = np.cumsum([f.shape[0] for f in feas])
    end_lab = np.cumsum([l.shape[0] for l in labs])
    return [names, np.concatenate(feas), np.concatenate(labs), np.asarray(end_fea), np.asarray(end_lab)]


def _read_features(rspec):
    rspec = str(rspec)
    if rspec.startswith("scp:"):
        return dict(read_mat_scp(rspec[4:]))
    if rspec.endswith(".scp"):
        return dict(read_mat_scp(rspec))
    return dict(read_mat_ark(rspec))


def _read_alignments(rspec):
    """One alignment ark (plain or .gz), or a folder holding ali*.gz (the reference's `gunzip -c folder/ali*.gz`)."""
    rspec = str(rspec)
    files = sorted(glob.glob(os.path.join(rspec, "ali*.gz"))) if os.path.isdir(rspec) else [rspec]
    if not files:
        raise IOError("no ali*.gz under %s" % rspec)
    out = {}
    for f in files:
        out.update(read_vec_int_ark(f))
    return out


def load_chunk(fea_rspec, lab_rspec, left, right, max_sequence_length, fea_only=False, device=None):
    """The reference's load_chunk (data_io.py:244-274) for tables that already exist on disk: `fea_rspec` is a feature
    scp ("key file:offset" lines) or ark, `lab_rspec` an integer-vector ark of pdf-ids (plain, .gz, or a folder of
    ali*.gz) - i.e. what `copy-feats scp:... ark:- | <fea_opts>` and `gunzip -c ali*.gz | ali-to-pdf ...` would deliver.
    Returns [data_name, data_set (float32, features + one label column), end_index_fea].  `device`: splice and
    normalise on that device (finish_chunk_device: the un-spliced features cross PCIe) and return data_set as a tensor
    resident there."""
    fea = _read_features(fea_rspec)
    lab = {} if fea_only else _read_alignments(lab_rspec)
    names, data_set, data_lab, end_index_fea, _ = load_dataset(fea, lab, max_sequence_length, fea_only)
    if device is not None:
        data_set, end_index_fea = finish_chunk_device(data_set, data_lab, end_index_fea, left, right, device)
    else:
        data_set, end_index_fea = finish_chunk(data_set, data_lab, end_index_fea, left, right)
    return [names, data_set, end_index_fea]


def _truth(v):
    v = str(v).strip().lower()
    if v in ("y", "yes", "t", "true", "on", "1"):
        return True
    if v in ("n", "no", "f", "false", "off", "0"):
        return False
    raise ValueError("invalid truth value %r" % v)


def dict_fea_lab_arch(config, fea_only):
    """utils.py:1889-1994: the feature streams, label sets and architectures the [model] lines actually use, each in
    order of first use.  fea_dict[name] = [name, fea_lst, fea_opts, cw_left, cw_right] (strings), lab_dict[name] =
    [name, lab_folder, lab_opts], arch_dict[name] = [section, name, seq_model]."""
    fea_field, lab_field = config["data_chunk"]["fea"], config["data_chunk"]["lab"]
    fea_names = re.findall("fea_name=(.*)\n", fea_field.replace(" ", ""))
    lab_names = re.findall("lab_name=(.*)\n", lab_field.replace(" ", ""))
    fea_dict, lab_dict, arch_dict = {}, {}, {}
    for line in config["model"]["model"].split("\n"):
        _, operation, inp1, inp2 = re.findall(r"(.*)=(.*)\((.*),(.*)\)", line)[0]
        for inp in (inp1, inp2):
            if inp in fea_names and inp not in fea_dict:
                pat = "fea_name=" + inp + "\nfea_lst=(.*)\nfea_opts=(.*)\ncw_left=(.*)\ncw_right=(.*)"
                fea_dict[inp] = (inp + "," + ",".join(re.findall(pat, fea_field)[0])).split(",")
        for inp in (inp1, inp2):
            if inp in lab_names and inp not in lab_dict and not fea_only:
                pat = "lab_name=" + inp + "\nlab_folder=(.*)\nlab_opts=(.*)"
                lab_dict[inp] = (inp + "," + ",".join(re.findall(pat, lab_field)[0])).split(",")
        if operation == "compute" and inp1 not in arch_dict:
            secs = [sec for sec in config.sections() if config[sec].get("arch_name") == inp1]
            if not secs:
                raise ValueError("no [architecture*] section has arch_name = %s" % inp1)
            arch_dict[inp1] = [secs[0], inp1, _truth(config[secs[0]]["arch_seq_model"])]
    return [fea_dict, lab_dict, arch_dict]


def read_lab_fea(cfg_file, fea_only, shared_list, output_folder=None):
    """The reference's chunk reader (data_io.py:536-655; `core.run_nn` runs it in a thread, core.py:492 / 511) for chunk
    cfg files whose feature pipelines are empty, i.e. whose tables already exist on disk: appends
    [data_name, data_end_index, fea_dict, lab_dict, arch_dict, data_set] to `shared_list`.

    Per (feature stream, label set) pair one load_chunk; streams with narrower context windows are trimmed to the widest
    one (:577-581); labels are taken from the first stream, streams are stacked left to right and get their column
    ranges appended to fea_dict (:590-592, :601-604), label columns follow (:622-627); non-sequence training chunks are
    shuffled with the global numpy RNG (:633-635).  `fea_opts` / `lab_opts` must be empty (or `lab_opts = pdf-ids`):
    a pipeline needs Kaldi and stays with the reference's reader.  data_set is float32 (the reference hands float64 to
    run_nn, which converts it).  PK_CHUNK_DEVICE=cuda: the chunk is finished ON the device (context windows spliced and
    the chunk normalised there, finish_chunk_device) and data_set is a tensor resident there - run_nn_dp's batch
    assembly gathers from it without a second copy."""
    dev = os.environ.get("PK_CHUNK_DEVICE") or None
    if dev == "cuda":  # "cuda" = THIS thread's current device (core.run_nn_dp binds its reader thread to the rank's GPU)
        dev = torch.device("cuda", torch.cuda.current_device())
    stack = (lambda cols: torch.cat([c if c.dim() == 2 else c[:, None] for c in cols], 1)) if dev else np.column_stack
    if not os.path.exists(cfg_file):
        raise IOError("The config file %s does not exist!" % cfg_file)
    config = configparser.ConfigParser()
    config.read(cfg_file)
    to_do = config["exp"]["to_do"]
    if to_do == "train":
        max_seq_length = int(config["batches"]["max_seq_length_train"])
    elif to_do == "valid":
        max_seq_length = int(config["batches"]["max_seq_length_valid"])
    else:
        max_seq_length = -1  # forward: sentences are never split
    fea_dict, lab_dict, arch_dict = dict_fea_lab_arch(config, fea_only)
    cw_left_max = max(int(fea_dict[f][3]) for f in fea_dict)
    cw_right_max = max(int(fea_dict[f][4]) for f in fea_dict)
    fea_index = 0
    data_set = labs = data_end_index = data_name = None
    for cnt_fea, fea in enumerate(fea_dict):
        fea_scp, fea_opts = fea_dict[fea][1], fea_dict[fea][2]
        cw_left, cw_right = int(fea_dict[fea][3]), int(fea_dict[fea][4])
        if fea_opts.strip():
            raise ValueError("read_lab_fea: fea_opts of %s is a Kaldi pipeline (%r); materialise it into an ark / scp "
                             "first or use the reference's reader" % (fea, fea_opts))
        if fea_only:
            lab_dict.update({"lab_name": "none"})
        for cnt_lab, lab in enumerate(lab_dict):
            lab_folder = None
            if not fea_only:
                lab_folder, lab_opts = lab_dict[lab][1], lab_dict[lab][2]
                if lab_opts.strip() not in ("", "pdf-ids"):
                    raise ValueError("read_lab_fea: lab_opts of %s is %r; store pdf-ids (ali-to-pdf run once) and set "
                                     "lab_opts = pdf-ids, or use the reference's reader" % (lab, lab_opts))
            name_fea, set_fea, end_fea = load_chunk(fea_scp, lab_folder, cw_left, cw_right, max_seq_length, fea_only, dev)
            lo, hi = cw_left_max - cw_left, set_fea.shape[0] - (cw_right_max - cw_right)
            labs_fea, set_fea = set_fea[lo:hi, -1], set_fea[lo:hi, 0:-1]
            end_fea = end_fea - lo
            end_fea[-1] = end_fea[-1] - (cw_right_max - cw_right)
            if cnt_fea == 0 and cnt_lab == 0:
                data_set, labs, data_end_index, data_name = set_fea, labs_fea, end_fea, name_fea
            else:
                if cnt_fea == 0:
                    labs = stack((labs, labs_fea))
                if cnt_lab == 0:
                    data_set = stack((data_set, set_fea))
                if data_name != name_fea:
                    raise ValueError("different sentence ids are detected for the different features. Please check "
                                     "again input feature lists")
                if not (data_end_index == end_fea).all():
                    raise ValueError("end_index must be the same for all the sentences")
            if cnt_lab == 0:
                fea_dict[fea].append(fea_index)
                fea_index = fea_index + set_fea.shape[1]
                fea_dict[fea].append(fea_index)
                fea_dict[fea].append(fea_dict[fea][6] - fea_dict[fea][5])
    if not fea_only:
        for cnt_lab, lab in enumerate(lab_dict):
            lab_dict[lab].append(data_set.shape[1] + cnt_lab)
    data_set = stack((data_set, labs))
    seq_model = any(arch_dict[a][2] for a in arch_dict)
    if not seq_model and to_do != "forward":
        if dev:  # np.random.shuffle draws the same swaps whatever the rows hold: shuffle an index with it, gather on the device
            perm = np.arange(data_set.shape[0])
            np.random.shuffle(perm)
            data_set = data_set[torch.as_tensor(perm, device=data_set.device)]
        else:
            np.random.shuffle(data_set)
    shared_list.extend([data_name, data_end_index, fea_dict, lab_dict, arch_dict, data_set])
