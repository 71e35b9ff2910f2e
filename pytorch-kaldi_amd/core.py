"""Data-parallel chunk function: ``run_nn_dp`` has the signature and return value of the reference's
``core.run_nn`` (core.py:439-753), so ``run_exp.py``, which resolves the chunk function by name from a module
called ``core`` (run_exp.py:129-131), can use it with ``run_nn_script = run_nn_dp`` once this module is importable
as ``core``.  What differs from the reference is everything SURVEY.md 8e / 8f-1..3 asks for:

* one process per GPU (``torch.distributed.run``): every rank walks the same chunk with the same RNG stream, takes
  its own columns of each batch (``dp.shard_batch``), and the gradient buckets are all-reduced over RCCL before the
  fused optimizer step; ``nn.DataParallel`` (which would split the TIME axis, core.py:103-104) is never used;
* batch assembly on the device: the reference copies sentence by sentence in a Python loop (core.py:581-598); here the
  loop only draws the random left padding (same ``random.randint`` sequence) and ONE gather builds the padded
  (T, B, D) batch from the resident chunk;
* fused flat-bucket optimizers whose state dicts stay torch.optim-compatible, so ``.pkl`` checkpoints written here
  load in the reference and vice versa;
* rank 0 alone writes the ``.pkl`` / ``.info`` / ``.ark`` files.

The chunk reader stays the reference's (Kaldi pipes, data_io.read_lab_fea): it is looked up as ``data_io.read_lab_fea``
on ``sys.path`` unless a ``reader`` callable with the same signature is passed in.
"""
import configparser
import os
import random
import struct
import sys
import threading
import time

import numpy as np
import torch

from . import _lib
from . import dp as _dp
from . import functional as F_
from . import nn as _nn
from ._lib import PkError
from .graphs import GraphedStep
from .optim import fused_optimizer_init
from .utils import forward_model, model_init, strtobool


def is_sequential_dict(config, arch_dict):
    """utils.py:2006-2014."""
    return any(strtobool(config[arch_dict[a][0]]["arch_seq_model"]) for a in arch_dict)


def load_counts(class_counts_file):
    """data_io.py:277-281 - one line ``[ c0 c1 ... ]``."""
    with open(class_counts_file) as f:
        row = next(f).strip().strip("[]").strip()
    return np.array([np.float32(v) for v in row.split()])


def write_mat(fd, m, key=""):
    """Binary Kaldi matrix record, byte-for-byte what data_io.write_mat emits (data_io.py:1200-1239):
    ``key SP \\0B (FM|DM) SP \\4 rows \\4 cols data``."""
    m = np.ascontiguousarray(m)
    if m.dtype == np.float32:
        tag = b"FM "
    elif m.dtype == np.float64:
        tag = b"DM "
    else:
        raise TypeError("write_mat: '%s', please use 'float32' or 'float64'" % m.dtype)
    if m.ndim != 2:
        raise ValueError("write_mat: expected a matrix, got %d dims" % m.ndim)
    if key != "":
        fd.write((key + " ").encode("latin1"))
    fd.write(b"\0B" + tag)
    fd.write(b"\x04" + struct.pack("<I", m.shape[0]) + b"\x04" + struct.pack("<I", m.shape[1]))
    fd.write(m.tobytes())


def sentence_lengths(data_end_index):
    """core.py:571-573: lengths from the cumulative end indices."""
    end = np.asarray(data_end_index, dtype=np.int64)
    return np.diff(end, prepend=0)


class BatchAssembler:
    """Zero-padded (max_len, B, D) batches with a random number of leading zeros per sentence
    (core.py:581-598), built with one gather from the chunk instead of B slice copies.  The chunk gets one
    extra all-zero row; the index map points padding at it."""

    def __init__(self, data_set, data_end_index, device):
        self.end = np.asarray(data_end_index, dtype=np.int64)
        self.len = sentence_lengths(data_end_index)
        self.beg = self.end - self.len
        self.n_rows = data_set.shape[0]
        pad = torch.zeros(1, data_set.shape[1], dtype=data_set.dtype, device=data_set.device)
        self.src = torch.cat((data_set, pad), 0)
        self.device = torch.empty(0, device=device).device  # normalised ("cuda" -> "cuda:<current>")
        self._ring = []  # pinned staging slots [buffer, event of its last H2D copy]
        self._slot = 0

    RING = 4

    def _draw(self, snt_index, batch_size, cols):
        """One ``random.randint`` per sentence of the WHOLE batch, in order, like the reference does (core.py:588),
        so that every rank (and the oracle) sees the same padding whatever columns it keeps."""
        lens = self.len[snt_index:snt_index + batch_size]
        max_len = int(lens.max())
        left = np.array([random.randint(0, int(max_len - n)) for n in lens], dtype=np.int64)
        cols = np.arange(batch_size) if cols is None else np.asarray(cols)
        return max_len, left[cols], lens[cols], self.beg[snt_index + cols]

    def index_map(self, snt_index, batch_size, cols=None):
        """(max_len, len(cols)) source-row map of the batch starting at sentence ``snt_index`` (host version)."""
        max_len, left, lens, beg = self._draw(snt_index, batch_size, cols)
        t = np.arange(max_len, dtype=np.int64)[:, None]
        rel = t - left[None, :]
        valid = (rel >= 0) & (rel < lens[None, :])
        idx = np.where(valid, beg[None, :] + rel, self.n_rows)
        return max_len, idx

    def _stage(self, host):
        """Asynchronous H2D copy of a small int64 array through a ring of pinned slots.  A slot is rewritten only
        after the event recorded behind its previous copy has completed: the host runs several batches ahead of the
        GPU (no sync inside a chunk), so an unguarded slot would be overwritten while its DMA is still pending."""
        n = host.numel()
        if len(self._ring) < self.RING:
            self._ring.append([torch.empty(max(n, 64), dtype=torch.int64).pin_memory(), None])
            slot = self._ring[-1]
        else:
            slot = self._ring[self._slot]
            self._slot = (self._slot + 1) % self.RING
            if slot[1] is not None:
                slot[1].synchronize()
            if slot[0].numel() < n:
                slot[0] = torch.empty(n, dtype=torch.int64).pin_memory()
        slot[0][:n].copy_(host.reshape(-1))
        dev = slot[0][:n].to(self.src.device, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        slot[1] = ev
        return dev.view(host.shape)

    def batch(self, snt_index, batch_size, cols=None):
        if self.src.is_cuda:
            # only (left, lens, beg) - 3 x B integers - cross PCIe; the (max_len, B) row map is built on the device
            max_len, left, lens, beg = self._draw(snt_index, batch_size, cols)
            llb = self._stage(torch.from_numpy(np.stack((left, lens, beg))))
            rel = torch.arange(max_len, device=self.src.device)[:, None] - llb[0][None, :]
            idx = torch.where((rel >= 0) & (rel < llb[1][None, :]), llb[2][None, :] + rel, self.n_rows)
            ncol = idx.shape[1]
            it = idx.reshape(-1)
        else:
            max_len, idx = self.index_map(snt_index, batch_size, cols)
            ncol = idx.shape[1]
            it = torch.from_numpy(idx.reshape(-1))
        out = self.src.index_select(0, it).view(max_len, ncol, -1)
        if out.device != self.device:
            if self.device.type == "cuda" and not out.is_cuda:
                out = out.pin_memory().to(self.device, non_blocking=True)
            else:  # (a chunk resident on another GPU: device-to-device copy; pin_memory() is for host tensors only)
                out = out.to(self.device)
        return max_len, out


def rank_columns(batch_size, rank, world):
    """Columns (sentences of a batch) this rank keeps: [r*B/N, (r+1)*B/N)."""
    if batch_size % world != 0:
        raise RuntimeError("batch size %d does not split over %d ranks" % (batch_size, world))
    local = batch_size // world
    return local, np.arange(rank * local, (rank + 1) * local)


def make_reducer(nns, optimizers, world, launch_bound=False, force=False):
    """Gradient all-reduce over the flat buckets of the fused optimizers (None on one rank).
    launch_bound: a recipe whose step is shorter than its own gradient exchange (MLP / SincNet on 128 frames: 0.3-4 ms
    per step, 27-40 MB of gradients): 4 MB buckets, so that the upper layers' gradients are on the wire while the lower
    layers still run backward, and the bf16 wire unless PK_DP_WIRE says otherwise (half the ring's payload; graded
    against the fp32 wire in tests/test_dp_gloo.py).  The collectives are part of the step's HIP graph there
    (round 5: RCCL kernels are capturable; the reference's only data-parallel hook is core.py:103-104, 537-538)."""
    if world <= 1 and not force:
        return None
    # bucket by bucket, behind the layer that produced it (PK_DP_OVERLAP=0: everything after backward)
    return _dp.GradReducer({k: nns[k] for k in nns}, flats={k: optimizers[k].flat for k in nns},
                           bucket_bytes=(4 << 20) if launch_bound else (8 << 20),
                           overlap=os.environ.get("PK_DP_OVERLAP", "1") != "0", force=force,
                           wire=default_wire(launch_bound))


def default_wire(launch_bound):
    """The format gradient buckets travel in.  PK_DP_WIRE decides when set.  Otherwise bf16 only where BOTH hold: the
    engine already computes with bf16 operands (PK_PRECISION=bf16) and the recipe is launch-bound.  In the parity mode
    (PK_PRECISION=fp32, the default) the wire stays fp32, so that N ranks equal the shard average the way the
    reference's DataParallel sum does (core.py:103-104)."""
    env = os.environ.get("PK_DP_WIRE")
    if env:
        return env
    return "bf16" if (launch_bound and F_.bf16_mode()) else "fp32"


def capture_on_every_rank(train_step, optimizers, inp, reducer, world):
    """Record the step as a HIP graph (graphs.GraphedStep) - or return None, on EVERY rank alike.  A capture that fails
    part-way through backward leaves the reducer with buckets handed over, half-counted signals and handles of work that
    was only recorded: it is reset before the eager step that follows.  The ranks agree on the outcome (all-reduce MIN)
    before anyone replays: a rank replaying captured collectives next to a rank issuing eager ones would hang."""
    graphed = None
    try:
        graphed = GraphedStep(train_step, [optimizers[k] for k in optimizers]).capture(inp)
    except (RuntimeError, PkError) as e:  # e.g. an Adam recipe: stay eager
        sys.stderr.write("run_nn_dp: HIP-graph capture not used (%s)\n" % (e,))
    ok = graphed is not None
    if world > 1:
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=inp.device)
        torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
        if ok and int(flag.item()) == 0:
            sys.stderr.write("run_nn_dp: HIP-graph capture failed on another rank: staying eager\n")
            ok = False
    if not ok:
        graphed = None
        if reducer is not None:
            reducer.reset()
    return graphed


def mean_over_ranks(loss_sum, err_sum, world):
    """Chunk loss / error of the whole job: mean of the per-rank sums (equal shards)."""
    if world <= 1:
        return loss_sum, err_sum
    both = torch.stack((loss_sum, err_sum))
    torch.distributed.all_reduce(both)
    return both[0] / world, both[1] / world


def _default_reader():
    """PK_READER=tables: the pipe-free reader of this package (data_io.read_lab_fea: chunk cfg files whose fea_opts are
    empty, i.e. tables that already exist on disk); otherwise the reference's reader with its Kaldi pipes."""
    if os.environ.get("PK_READER", "reference") == "tables":
        from . import data_io as pk_io
        return pk_io.read_lab_fea
    try:
        import data_io  # the reference's, when running inside a PyTorch-Kaldi checkout
    except ImportError as e:
        raise RuntimeError("run_nn_dp needs a chunk reader: put PyTorch-Kaldi's data_io.py on sys.path, set "
                           "PK_READER=tables, or pass reader=callable(cfg_file, is_production, shared_list, "
                           "output_folder)") from e
    return data_io.read_lab_fea


def _on_device(read, device):
    """The reader as a thread target that first binds the thread to this rank's GPU: the HIP current device is
    THREAD-local and a fresh thread starts on device 0, so without this every rank of a node would finish its chunk
    (PK_CHUNK_DEVICE=cuda: 1.9 GB fp32 + temporaries) on GPU 0."""
    def run(*args):
        if device.type == "cuda":
            torch.cuda.set_device(device)
        return read(*args)
    return run


def _to_tensor(data_set, save_gpumem, use_cuda):
    t = torch.from_numpy(data_set).float() if isinstance(data_set, np.ndarray) else data_set.float()
    return t.cuda() if (use_cuda and not save_gpumem) else t


def run_nn_dp(data_name, data_set, data_end_index, fea_dict, lab_dict, arch_dict, cfg_file, processed_first,
              next_config_file, reader=None):
    """Process one chunk as ``[exp] to_do`` says (train / valid / forward) and return the next chunk's
    ``[data_name, data_set, data_end_index, fea_dict, lab_dict, arch_dict]`` (core.py:753)."""
    if not os.path.exists(cfg_file):
        sys.stderr.write("ERROR: The config file %s does not exist!\n" % (cfg_file))
        sys.exit(0)
    config = configparser.ConfigParser()
    config.read(cfg_file)
    seed = int(config["exp"]["seed"])
    _nn.drain_mask_prefetch()  # (PK_MASK_RNG=reference: no helper thread may be drawing while the generator is re-seeded)
    torch.manual_seed(seed)
    random.seed(seed)
    np.random.seed(seed)

    exp = config["exp"]
    output_folder = exp["out_folder"]
    use_cuda = strtobool(exp["use_cuda"])
    if not use_cuda:
        raise RuntimeError("run_nn_dp drives the MI355X engine: it needs use_cuda=True (there is no CPU path)")
    to_do = exp["to_do"]
    info_file = exp["out_info"]
    model = config["model"]["model"].split("\n")
    forward_outs = config["forward"]["forward_out"].split(",")
    forward_normalize_post = list(map(strtobool, config["forward"]["normalize_posteriors"].split(",")))
    forward_count_files = config["forward"]["normalize_with_counts_from"].split(",")
    require_decodings = list(map(strtobool, config["forward"]["require_decoding"].split(",")))
    save_gpumem = strtobool(exp["save_gpumem"])
    is_production = strtobool(exp["production"]) if "production" in exp else 0
    batch_size = {"train": lambda: int(config["batches"]["batch_size_train"]),
                  "valid": lambda: int(config["batches"]["batch_size_valid"]), "forward": lambda: 1}[to_do]()

    rank, world, local_rank = _dp.init_from_env()
    device = torch.device("cuda", torch.cuda.current_device())
    read = reader if reader is not None else _default_reader()

    if processed_first:
        shared_list = []
        p = threading.Thread(target=_on_device(read, device), args=(cfg_file, is_production, shared_list, output_folder))
        p.start()
        p.join()
        data_name, data_end_index, fea_dict, lab_dict, arch_dict, data_set = shared_list[:6]
        data_set = _to_tensor(data_set, save_gpumem, use_cuda)
    shared_list = []
    p = threading.Thread(target=_on_device(read, device), args=(next_config_file, is_production, shared_list, output_folder))
    p.start()

    inp_out_dict = fea_dict
    nns, costs = model_init(inp_out_dict, model, config, arch_dict, use_cuda, False, to_do)
    optimizers = fused_optimizer_init(nns, config, arch_dict)
    for o_ in optimizers.values():
        o_.zero_in_step = True  # this loop zeroes the gradients in front of every backward pass and never reads them behind step()
    for net in nns.keys():
        pt_file_arch = config[arch_dict[net][0]]["arch_pretrain_file"]
        if pt_file_arch != "none":
            checkpoint_load = torch.load(pt_file_arch, map_location=device)
            nns[net].load_state_dict(checkpoint_load["model_par"])
            optimizers[net].load_state_dict(checkpoint_load["optimizer_par"])
            optimizers[net].param_groups[0]["lr"] = float(config[arch_dict[net][0]]["arch_lr"])
    seq_model = is_sequential_dict(config, arch_dict)
    reducer = make_reducer(nns, optimizers, world, launch_bound=not seq_model) if to_do == "train" else None

    post_file = {}
    if to_do == "forward" and rank == 0:
        for out_id, name in enumerate(forward_outs):
            suffix = "_to_decode.ark" if require_decodings[out_id] else ".ark"
            post_file[name] = open(info_file.replace(".info", "_" + name + suffix), "wb")

    if seq_model or to_do == "forward":
        N_batches = int(len(data_name) / batch_size)
    else:
        N_batches = int(data_set.shape[0] / batch_size)
    end = np.asarray(data_end_index, dtype=np.int64)
    assembler = BatchAssembler(data_set, data_end_index, device) if seq_model else None
    local, cols = rank_columns(batch_size, rank, world) if to_do != "forward" else (1, np.arange(1))
    counts = {i: load_counts(forward_count_files[i]) for i in range(len(forward_outs))
              if to_do == "forward" and forward_normalize_post[i]}

    def train_step(inp_):
        # (a training step - forward and the backward pass that accumulates into .grad: kernels may add to the flat
        # .grad themselves, and recurrent layers take their BatchNorm affine as views of the flat buffer)
        with F_.accumulating_backward():
            outs = forward_model(fea_dict, lab_dict, arch_dict, model, nns, costs, inp_, inp_out_dict, 0 if not seq_model
                                 else inp_.shape[0], local, to_do, forward_outs)
            for opt in optimizers.keys():
                optimizers[opt].zero_grad()
            outs["loss_final"].backward()
        if reducer is not None:
            reducer.finish()
        for opt in optimizers.keys():
            if not strtobool(config[arch_dict[opt][0]]["arch_freeze"]):
                optimizers[opt].step()
        return {"loss_final": outs["loss_final"].detach(), "err_final": outs["err_final"].detach()}

    # fixed-shape (non-sequence) training batches are launch-bound: after a few eager batches the whole step - on
    # several ranks with its bucketed RCCL all-reduces, which the first eager batches have taught the reducer to place -
    # is replayed as one HIP graph (graphs.py); PK_HIPGRAPH=0 keeps it eager
    GRAPH_WARMUP = 3
    graph_ok = (to_do == "train" and not seq_model and os.environ.get("PK_HIPGRAPH", "1") != "0"
                and N_batches > GRAPH_WARMUP + 1)
    if graph_ok and world > 1 and torch.distributed.get_backend() != "nccl":
        graph_ok = False  # only RCCL's collectives are stream work a capture can record (gloo synchronises on the host)
    graphed = None
    if reducer is not None and rank == 0:
        sys.stderr.write("run_nn_dp: %d ranks, %d gradient buckets, %s wire\n" % (world, len(reducer.buckets), reducer.wire))

    start_time = time.time()
    loss_sum = torch.zeros((), device=device)
    err_sum = torch.zeros((), device=device)
    snt_index, beg_snt = 0, 0
    fence = F_.StepFence()  # eager steps: at most PK_STEPS_IN_FLIGHT of them enqueued ahead of the GPU
    for i in range(N_batches):
        max_len = 0
        if to_do == "forward":
            if rank != 0:
                continue
            snt_len = int(end[snt_index]) - beg_snt
            inp = data_set[beg_snt:beg_snt + snt_len, :].contiguous().to(device)
            beg_snt = int(end[snt_index])
            snt_index += 1
            if seq_model:
                max_len, inp = snt_len, inp.view(snt_len, 1, -1)
        elif seq_model:
            max_len, inp = assembler.batch(snt_index, batch_size, cols)
            snt_index += batch_size
        else:
            b0 = i * batch_size + rank * local
            inp = data_set[b0:b0 + local, :].contiguous().to(device, non_blocking=True)
        nb = local if to_do != "forward" else 1
        if to_do == "train":
            if graphed is not None:
                outs_dict = graphed(inp)
            elif graph_ok and i == GRAPH_WARMUP:
                graphed = capture_on_every_rank(train_step, optimizers, inp, reducer, world)
                graph_ok = graphed is not None
                outs_dict = graphed(inp) if graphed is not None else train_step(inp)
            else:
                outs_dict = train_step(inp)
            if graphed is None:
                fence()
        else:
            with torch.no_grad():
                outs_dict = forward_model(fea_dict, lab_dict, arch_dict, model, nns, costs, inp, inp_out_dict, max_len, nb,
                                          to_do, forward_outs)
        if to_do == "forward":
            for out_id, name in enumerate(forward_outs):
                out_save = outs_dict[name].detach().cpu().numpy()
                if forward_normalize_post[out_id]:
                    c = counts[out_id]
                    out_save = out_save - np.log(c / np.sum(c))
                write_mat(post_file[name], out_save, data_name[i])
        else:
            loss_sum += outs_dict["loss_final"].detach()
            err_sum += outs_dict["err_final"].detach()
    # the only full host sync of the chunk (the reference syncs once per batch for its progress bar, core.py:689; eager
    # steps wait for the step four back - functional.StepFence)
    if to_do != "forward":
        loss_sum, err_sum = mean_over_ranks(loss_sum, err_sum, world)
    torch.cuda.synchronize()
    # a bounded-spin time-out of a persistent recurrent kernel invalidates that launch's results: it must surface
    # BEFORE this chunk's .pkl / .info are written (the counter is host-mapped, read after the sync above)
    _lib.raise_if_persist_failed()
    F_.raise_if_bad_labels()
    elapsed_time_chunk = time.time() - start_time
    _nn.drain_mask_prefetch()  # masks drawn ahead for a batch that will not come: back into the generator
    loss_tot = loss_sum / max(N_batches, 1)
    err_tot = err_sum / max(N_batches, 1)

    if to_do == "train" and rank == 0:
        for net in nns.keys():
            checkpoint = {"model_par": nns[net].state_dict(), "optimizer_par": optimizers[net].state_dict()}
            torch.save(checkpoint, info_file.replace(".info", "_" + arch_dict[net][0] + ".pkl"))
    for f in post_file.values():
        f.close()
    if rank == 0:
        with open(info_file, "w") as text_file:
            text_file.write("[results]\n")
            if to_do != "forward":
                text_file.write("loss=%s\n" % loss_tot.cpu().numpy())
                text_file.write("err=%s\n" % err_tot.cpu().numpy())
            text_file.write("elapsed_time_chunk=%f\n" % elapsed_time_chunk)
    if world > 1:
        torch.distributed.barrier()

    p.join()
    if len(shared_list) < 6:
        return [None, None, None, None, None, None]
    data_name, data_end_index, fea_dict, lab_dict, arch_dict, data_set = shared_list[:6]
    return [data_name, _to_tensor(data_set, save_gpumem, use_cuda), data_end_index, fea_dict, lab_dict, arch_dict]


run_nn = run_nn_dp  # so that cfgs with run_nn_script=run_nn pick the data-parallel loop up when this module is `core`
