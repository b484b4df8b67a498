"""Flat, device-resident graph dataset and device-side batch assembly.

Replaces, for the hot path, the reference's pickled PyG InMemoryDataset + python collate
(/root/reference/matdeeplearn/process/process.py:133-153,520-523 and the PyG DataLoader built at
/root/reference/matdeeplearn/training/training.py:300-325): the whole dataset lives in HBM as
flat arrays; a batch is assembled ON THE DEVICE from a list of graph ids, with the edges already
in CSR-by-target order, and the 50-wide Gaussian edge features are expanded on the fly by the K1
HIP kernel from the stored normalised distance (4 B/edge instead of 200 B/edge at rest).

Batch fields match what the reference models read from a PyG `Batch`
(SURVEY 8b): x, edge_index, edge_attr, edge_weight, batch, u, y (+ num_graphs, csr).
"""
import numpy as np
import torch

from .. import ops
from . import graph as pg


class GraphRecord:
    """What `dataset[i]` exposes to model constructors (cgcnn.py:58-61, megnet.py:229)."""

    def __init__(self, y, u):
        self.y, self.u = y, u


def _weak_call(obj, method):
    """obj.method() through a weak reference (None once obj is gone): what the segment-index registry may keep without
    keeping a batch's index tensors alive."""
    import weakref
    ref = weakref.ref(obj)

    def call():
        o = ref()
        return None if o is None else getattr(o, method)()
    return call


class Batch:
    """Batched graphs; `edge_index` ([2,E] int64, PyG convention) is materialised lazily because the
    product kernels consume `csr` (int32) directly."""

    def __init__(self, **kw):
        self.__dict__.update(kw)
        self._edge_index = None

    @property
    def edge_index(self):
        fill = self.__dict__.get("_edge_index_fill")
        if fill is not None and self.__dict__.get("_edge_index_stale", False):
            fill()                                                        # (static batch: refill the int64 view on first use)
            self._edge_index_stale = False
        if self._edge_index is None:
            self._edge_index = torch.stack([self.csr.src.long(), self.csr.tgt.long()])
            ops.register_csr(self._edge_index, self.csr)
            if self.csr.eperm is None and self._edge_index.is_cuda:       # scatter(..., edge_index[k]) without a sort
                ops.register_seg_index(self._edge_index[0], _weak_call(self.csr, "seg_src"), owner=self._edge_index)
                ops.register_seg_index(self._edge_index[1], _weak_call(self.csr, "seg_tgt"), owner=self._edge_index)
        return self._edge_index

    def to(self, device):
        return self  # already resident; kept for `data.to(rank)` call-site compatibility (training.py:39)

    def tensors(self):
        """every device tensor the batch owns (features, targets, index arrays of the CSR)"""
        out = [v for v in self.__dict__.values() if torch.is_tensor(v)]
        csr = self.__dict__.get("csr")
        if csr is not None:
            out += [t for t in (csr.rowptr, csr.src, csr.tgt, csr.eperm) if torch.is_tensor(t)]
        return out

    def record_stream(self, stream):
        """Tell the caching allocator that `stream` uses this batch's memory.  A batch is OWNED by the stream that was current
        when it was collated / taken from the loader (DeviceLoader, take_ahead); a consumer that touches it on ANOTHER stream —
        a data-parallel side stream, a user stream in predict() — calls this once (and orders the streams with an event), or
        the memory may be handed out again while that stream still reads it once the batch is dropped."""
        if stream is None:
            return self
        for t in self.tensors():
            if t.is_cuda:
                t.record_stream(stream)
        return self


class GraphDataset:
    """Flat arrays (numpy on the host until .to(device)):
      node_ptr/edge_ptr [Gn+1]; x [Nt,F]; z [Nt]; in_deg [Nt]; src/tgt [Et] graph-local int32,
      edges sorted by target inside every graph; dist [Et] raw distance (edge_weight);
      dist_norm [Et] min/max-normalised over the dataset (process.py:626-653); y [Gn,T]; ids."""

    def __init__(self, node_ptr, edge_ptr, x, z, src, tgt, dist, y, ids, num_edge_features=50, dist_range=None):
        self.node_ptr = np.asarray(node_ptr, dtype=np.int64)
        self.edge_ptr = np.asarray(edge_ptr, dtype=np.int64)
        self.x, self.z = x, z
        self.src, self.tgt, self.dist = src, tgt, dist
        self.y = np.asarray(y, dtype=np.float32).reshape(len(self.node_ptr) - 1, -1)
        self.ids = list(ids)
        self.num_edge_features = int(num_edge_features)
        self.target_index = 0
        self._side = None               # side stream of collate_ahead()
        lo, hi = dist_range if dist_range is not None else (float(dist.min()), float(dist.max()))
        self.dist_range = (lo, hi)
        # same fp32 arithmetic as NormalizeEdge (process.py:650-653)
        self.dist_norm = ((torch.from_numpy(np.asarray(dist, dtype=np.float32)) - np.float32(lo))
                          / (np.float32(hi) - np.float32(lo))).numpy()
        gl_tgt = np.asarray(tgt, dtype=np.int64) + np.repeat(self.node_ptr[:-1], np.diff(self.edge_ptr))
        self.in_deg = np.bincount(gl_tgt, minlength=len(z)).astype(np.int32)
        # exclusive in-degree prefix inside every graph: batch rowptr[n] = (first batch edge of the graph) + lrowptr
        csum = np.concatenate([[0], np.cumsum(self.in_deg, dtype=np.int64)])
        nodes_per_graph = np.diff(self.node_ptr)
        self.lrowptr = (csum[:-1] - np.repeat(csum[self.node_ptr[:-1]], nodes_per_graph)).astype(np.int32)
        self.device = None
        self._dev = {}

    # ---- reference-facing surface -------------------------------------------------------------
    def __len__(self):
        return len(self.node_ptr) - 1

    @property
    def num_features(self):
        return int(self.x.shape[1])

    def __getitem__(self, i):
        yi = torch.from_numpy(self.y[i])
        y = yi[self.target_index] if self.target_index != -1 else yi.view(1, -1)  # GetY, process.py:695-703
        return GraphRecord(y=y, u=torch.zeros(1, 3))

    @property
    def num_nodes(self):
        return int(self.node_ptr[-1])

    @property
    def num_edges(self):
        return int(self.edge_ptr[-1])

    # ---- flat on-disk format (replaces torch.save((data, slices)) of process.py:520-532) ---------------
    FLAT_MAGIC = b"MDLFLAT1"
    _FLAT_FIELDS = ("node_ptr", "edge_ptr", "x", "z", "src", "tgt", "dist", "y")

    def save_flat(self, path):
        """One file: 8-byte magic, 8-byte little-endian header length, JSON header (dtype / shape / byte offset of every
        array, edge-feature count, distance range, target index, structure ids), then the raw arrays, each 64-byte
        aligned — the arrays of the in-memory layout as they are (graph-local CSR-by-target indices, 4 bytes of
        distance per edge; the 50-wide RBF features are expanded on the device per batch), so loading is a memory map."""
        import json
        arrays = {k: np.ascontiguousarray(getattr(self, k)) for k in self._FLAT_FIELDS}
        meta, off = {}, 0
        for k, a in arrays.items():
            off = -(-off // 64) * 64
            meta[k] = {"dtype": a.dtype.str, "shape": list(a.shape), "offset": off}
            off += a.nbytes
        header = json.dumps({"arrays": meta, "num_edge_features": self.num_edge_features, "dist_range": list(self.dist_range),
                             "target_index": int(self.target_index), "ids": [str(i) for i in self.ids]}).encode()
        header += b" " * (-(16 + len(header)) % 64)
        with open(path, "wb") as f:
            f.write(self.FLAT_MAGIC)
            f.write(len(header).to_bytes(8, "little"))
            f.write(header)
            base = f.tell()
            for k, a in arrays.items():
                f.seek(base + meta[k]["offset"])
                f.write(a.tobytes())
        return path

    @classmethod
    def load_flat(cls, path, mmap=True):
        """Inverse of save_flat.  With mmap=True the arrays are read-only views of the file (pages fault in on first use;
        .to(device) streams them to HBM); derived arrays (normalised distance, in-degree prefix) are recomputed."""
        import json
        with open(path, "rb") as f:
            if f.read(8) != cls.FLAT_MAGIC:
                raise ValueError("%s is not a matdeeplearn_amd flat dataset" % path)
            hlen = int.from_bytes(f.read(8), "little")
            hdr = json.loads(f.read(hlen).decode())
            base = 16 + hlen
        arr = {}
        for k, m in hdr["arrays"].items():
            shape = tuple(m["shape"])
            if mmap:
                arr[k] = np.memmap(path, dtype=np.dtype(m["dtype"]), mode="r", offset=base + m["offset"], shape=shape)
            else:
                with open(path, "rb") as f:
                    f.seek(base + m["offset"])
                    n = int(np.prod(shape)) if shape else 1
                    arr[k] = np.fromfile(f, dtype=np.dtype(m["dtype"]), count=n).reshape(shape)
        ds = cls(arr["node_ptr"], arr["edge_ptr"], arr["x"], arr["z"], arr["src"], arr["tgt"], arr["dist"], arr["y"],
                 hdr["ids"], hdr["num_edge_features"], tuple(hdr["dist_range"]))
        ds.target_index = hdr["target_index"]
        return ds

    # ---- device residency -----------------------------------------------------------------------
    def to(self, device):
        device = torch.device(device)
        # File-backed (memory-mapped) arrays go through an anonymous copy: the runtime pins the pages of a large pageable
        # source for the upload, and pinned PAGE-CACHE pages (of a flat file written seconds ago: writeback, reclaim) later
        # trigger MMU-notifier invalidations that evict the process's GPU queues — measured as a 100-200 ms window, 2-3 s
        # after start, in which every kernel takes 2-4x longer (bench runs reading the dataset cache: 12 ms steps).
        def f(a, dt=None):
            a = np.array(a) if isinstance(a, np.memmap) else np.ascontiguousarray(a)
            return torch.as_tensor(a).to(device=device, dtype=dt)
        self._dev = dict(node_ptr=f(self.node_ptr), edge_ptr=f(self.edge_ptr), x=f(self.x, torch.float32),
                         src=f(self.src, torch.int32), tgt=f(self.tgt, torch.int32),
                         dist=f(self.dist, torch.float32), dist_norm=f(self.dist_norm, torch.float32),
                         in_deg=f(self.in_deg, torch.int32), lrowptr=f(self.lrowptr, torch.int32),
                         y=f(self.y, torch.float32),
                         offsets=ops.rbf_offsets(0.0, 1.0, self.num_edge_features, device))
        self.device = device
        return self

    def by_source(self):
        """Per-graph by-SOURCE order of the edges (device tensors, computed once): eperm_s [Et] = graph-local edge id at every
        by-source position (stable: by target inside a source), lrowptr_s [Nt] = exclusive out-degree prefix inside the
        node's graph.  The batch assembly turns them into the transposed CSR without a per-batch sort."""
        if "eperm_s" not in self._dev:
            epg = np.diff(self.edge_ptr)
            gl_src = np.asarray(self.src, dtype=np.int64) + np.repeat(self.node_ptr[:-1], epg)
            perm = np.argsort(gl_src, kind="stable")                        # keys are graph-monotone: blocks stay in place
            eperm_s = (perm - np.repeat(self.edge_ptr[:-1], epg)).astype(np.int32)
            out_deg = np.bincount(gl_src, minlength=len(self.z))
            csum = np.concatenate([[0], np.cumsum(out_deg, dtype=np.int64)])
            lrowptr_s = (csum[:-1] - np.repeat(csum[self.node_ptr[:-1]], np.diff(self.node_ptr))).astype(np.int32)
            self._dev["eperm_s"] = torch.from_numpy(eperm_s).to(self.device)
            self._dev["lrowptr_s"] = torch.from_numpy(lrowptr_s).to(self.device)
        return self._dev["eperm_s"], self._dev["lrowptr_s"]

    def _upload(self, arr, dev, ring=8):
        """host int64 array -> device tensor through a ring of pinned staging buffers (async; a slot is reused only after the
        copy that read it has finished)"""
        st = self.__dict__.setdefault("_pin", {"bufs": [None] * ring, "evs": [None] * ring, "k": 0})
        k = st["k"]
        st["k"] = (k + 1) % ring
        n = int(arr.shape[0])
        buf = st["bufs"][k]
        if buf is None or buf.numel() < n:
            buf = st["bufs"][k] = torch.empty(max(n, 1024), dtype=torch.int64).pin_memory()
        elif st["evs"][k] is not None:
            st["evs"][k].synchronize()
        buf[:n].copy_(torch.from_numpy(arr))
        out = torch.empty(n, dtype=torch.int64, device=dev)
        out.copy_(buf[:n], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))
        st["evs"][k] = ev
        return out

    def _zero_u(self, B, dev):
        """the reference's global state input: zeros [B, 3] (process.py:365) — constant, so one tensor per batch size"""
        c = self.__dict__.setdefault("_u0", {})
        t = c.get((B, str(dev)))
        if t is None:
            t = c[(B, str(dev))] = torch.zeros(B, 3, device=dev)
            if t.is_cuda:
                # a cached constant is shared by every stream that ever collates (the prefetching side stream creates it as
                # often as not): wait once for its fill, after which no consumer needs an event; it is never freed
                torch.cuda.current_stream(t.device).synchronize()
        return t

    def assemble_hip(self, ids, x_dtype=torch.float32):
        """K8: the whole batch assembly in ONE HIP launch (one workgroup per graph).  Prefix offsets are
        computed on the host from node_ptr/edge_ptr (B numbers) and uploaded with the ids in one copy."""
        from .. import _lib
        d = self._dev
        ids = np.asarray(ids, dtype=np.int64)
        B = len(ids)
        ncnt = self.node_ptr[ids + 1] - self.node_ptr[ids]
        ecnt = self.edge_ptr[ids + 1] - self.edge_ptr[ids]
        noff = np.concatenate([[0], np.cumsum(ncnt)])
        eoff = np.concatenate([[0], np.cumsum(ecnt)])
        N, E = int(noff[-1]), int(eoff[-1])
        dev = self.device
        # one async copy from a ring of pinned buffers (a pageable source is staged through several blit kernels that sit
        # in front of the assembly kernel on the stream: 4 x 4.8 us per step in the bench trace)
        pack = self._upload(np.concatenate([ids, noff, eoff]).astype(np.int64), dev)
        ids_d, noff_d, eoff_d = pack[:B], pack[B:2 * B + 1], pack[2 * B + 1:]
        F = self.num_features
        x = torch.empty((N, F), dtype=x_dtype, device=dev)
        batch = torch.empty(N, dtype=torch.int64, device=dev)
        rowptr = torch.empty(N + 1, dtype=torch.int32, device=dev)
        src = torch.empty(E, dtype=torch.int32, device=dev)
        tgt = torch.empty(E, dtype=torch.int32, device=dev)
        ew = torch.empty(E, dtype=torch.float32, device=dev)
        dn = torch.empty(E, dtype=torch.float32, device=dev)
        y = torch.empty(B, dtype=torch.float32, device=dev)
        p = _lib.ptr
        _lib.check(_lib.lib().mdl_assemble_batch(
            p(ids_d), p(noff_d), p(eoff_d), p(d["node_ptr"]), p(d["edge_ptr"]), p(d["x"]), p(d["src"]), p(d["tgt"]),
            p(d["dist"]), p(d["dist_norm"]), p(d["lrowptr"]), p(d["y"]), p(x), p(batch), p(rowptr), p(src), p(tgt),
            p(ew), p(dn), p(y), B, F, self.y.shape[1], int(self.target_index), _lib.dtype_code(x), _lib.stream()),
            "mdl_assemble_batch")
        csr = ops.EdgeCSR(rowptr, src, tgt, None, N, E)
        # the by-source index comes from the dataset as well (built on first use by the operators that need it)
        def transposed():
            eperm_s, lrowptr_s = self.by_source()
            rowptr_s = torch.empty(N + 1, dtype=torch.int32, device=dev)
            col_s, eid_s, src_s = (torch.empty(E, dtype=torch.int32, device=dev) for _ in range(3))
            _lib.check(_lib.lib().mdl_assemble_transposed(
                p(ids_d), p(noff_d), p(eoff_d), p(d["node_ptr"]), p(d["edge_ptr"]), p(d["src"]), p(d["tgt"]), p(eperm_s),
                p(lrowptr_s), p(rowptr_s), p(col_s), p(eid_s), p(src_s), B, -1, _lib.stream()), "mdl_assemble_transposed")
            return rowptr_s, col_s, eid_s, src_s
        csr.set_transposed_builder(transposed)
        ops.register_seg_index(src, _weak_call(csr, "seg_src"))   # scatter(..., csr.row / csr.col): no per-batch sort
        ops.register_seg_index(tgt, _weak_call(csr, "seg_tgt"))
        # (_pack: the uploaded ids / offsets the lazily built by-source index reads later, possibly on another stream than the one
        # this batch was assembled on — Batch.tensors() must see every allocation of the assembly, see take_ahead)
        return Batch(x=x, edge_attr=None, edge_weight=ew, batch=batch, y=y, u=self._zero_u(B, dev),
                     num_graphs=B, csr=csr, num_nodes=N, num_edges=E,
                     structure_id=[self.ids[i] for i in ids], _pack=pack), dn

    def collate(self, ids, edge_dtype=torch.float32, rbf=None, x_dtype=None):
        """Assemble the batch for graph ids (host int array) on the device and expand the edge
        features with the K1 HIP kernel.  No host sync: sizes come from the host node/edge_ptr.
        `rbf` lets the CPU test-suite inject the oracle expansion; the product default is the kernel."""
        if rbf is None and self.device is not None and self.device.type == "cuda" and self.target_index != -1:
            b, dist_norm = self.assemble_hip(ids, x_dtype if x_dtype is not None else torch.float32)
        else:
            b, dist_norm = self.assemble(ids)
        if rbf is None:
            b.edge_attr = ops.rbf_expand(dist_norm, 0.0, 1.0, self.num_edge_features, 0.2, out_dtype=edge_dtype,
                                         offsets=self._dev["offsets"])
        else:
            b.edge_attr = rbf(dist_norm).to(edge_dtype)
        return b

    def collate_ahead(self, ids, edge_dtype=torch.float32, x_dtype=None):
        """collate() of a LATER batch on a side stream: the assembly (K8) and the RBF expansion (K1) are short, latency-bound
        launches that fit beside the conv kernels of the step that is running (its tails leave CUs idle), which is what the
        worker processes of the reference's DataLoader (training.py:300-325) buy on the host.  Returns a handle for
        take_ahead(); the caller's stream is not touched."""
        dev = self.device
        if self._side is None:
            self._side = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(self._side):
            b = self.collate(ids, edge_dtype=edge_dtype, x_dtype=x_dtype)
            ev = torch.cuda.Event()
            ev.record(self._side)
        return b, ev

    def take_ahead(self, handle):
        """the batch of a collate_ahead() handle, ready for the CURRENT stream: that stream waits for the side stream's
        launches, and the caching allocator is told that the batch's memory (allocated on the side stream) is in use here"""
        b, ev = handle
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(ev)
        return b.record_stream(cur)       # (from here on the batch belongs to `cur`: see Batch.record_stream)

    def assemble(self, ids):
        """Index arithmetic of the batch assembly (plain tensor ops; device agnostic so the CPU test
        suite can check it against a per-graph concatenation).  Returns (Batch without edge_attr,
        normalised distances [E])."""
        if self.device is None:
            raise ops.MdlError("GraphDataset.collate: call .to(device) first")
        d = self._dev
        ids = np.asarray(ids, dtype=np.int64)
        B = len(ids)
        ncnt = self.node_ptr[ids + 1] - self.node_ptr[ids]
        ecnt = self.edge_ptr[ids + 1] - self.edge_ptr[ids]
        N, E = int(ncnt.sum()), int(ecnt.sum())
        dev = self.device
        ids_d = torch.from_numpy(ids).to(dev, non_blocking=True)
        ncnt_d = torch.from_numpy(ncnt).to(dev, non_blocking=True)
        ecnt_d = torch.from_numpy(ecnt).to(dev, non_blocking=True)
        noff = torch.cumsum(ncnt_d, 0) - ncnt_d            # first batch node of every graph
        eoff = torch.cumsum(ecnt_d, 0) - ecnt_d
        ar = torch.arange(B, device=dev)
        gon = torch.repeat_interleave(ar, ncnt_d, output_size=N)   # graph of node  (= `batch`)
        goe = torch.repeat_interleave(ar, ecnt_d, output_size=E)   # graph of edge
        nsrc = d["node_ptr"].index_select(0, ids_d).index_select(0, gon) + (torch.arange(N, device=dev) - noff.index_select(0, gon))
        esrc = d["edge_ptr"].index_select(0, ids_d).index_select(0, goe) + (torch.arange(E, device=dev) - eoff.index_select(0, goe))
        shift = noff.index_select(0, goe).to(torch.int32)
        src = d["src"].index_select(0, esrc) + shift
        tgt = d["tgt"].index_select(0, esrc) + shift
        rowptr = torch.zeros(N + 1, dtype=torch.int32, device=dev)
        rowptr[1:] = torch.cumsum(d["in_deg"].index_select(0, nsrc), 0, dtype=torch.int32)
        csr = ops.EdgeCSR(rowptr, src, tgt, None, N, E)
        y = d["y"].index_select(0, ids_d)
        y = y[:, self.target_index] if self.target_index != -1 else y
        return Batch(x=d["x"].index_select(0, nsrc), edge_attr=None,
                     edge_weight=d["dist"].index_select(0, esrc), batch=gon, y=y,
                     u=torch.zeros(B, 3, device=dev), num_graphs=B, csr=csr, num_nodes=N, num_edges=E,
                     structure_id=[self.ids[i] for i in ids]), d["dist_norm"].index_select(0, esrc)


# ------------------------------------------------------------------------------------------------
# builders
# ------------------------------------------------------------------------------------------------
def from_graphs(graphs, ys, ids, num_edge_features=50, dist_range=None):
    """graphs: list of dict(x, edge_index (reference order), edge_weight) from graph.build_graph."""
    node_ptr, edge_ptr = [0], [0]
    xs, zs, srcs, tgts, dists = [], [], [], [], []
    for g in graphs:
        ei, ew, _ = pg.sort_by_target(g["edge_index"], g["edge_weight"])
        n = g["x"].shape[0]
        node_ptr.append(node_ptr[-1] + n)
        edge_ptr.append(edge_ptr[-1] + ei.shape[1])
        xs.append(g["x"])
        zs.append(g.get("z", np.zeros(n, dtype=np.int64)))
        srcs.append(ei[0].astype(np.int32))
        tgts.append(ei[1].astype(np.int32))
        dists.append(ew.astype(np.float32))
    return GraphDataset(node_ptr, edge_ptr, np.concatenate(xs).astype(np.float32), np.concatenate(zs),
                        np.concatenate(srcs), np.concatenate(tgts), np.concatenate(dists), ys, ids,
                        num_edge_features, dist_range)


def from_structures(structs, ys, ids, radius=8.0, max_neighbors=12, num_edge_features=50, dictionary=None):
    """structs: iterable of dict(positions, numbers, cell, pbc) — e.g. graph.read_ase_json outputs."""
    graphs = []
    for s in structs:
        g = pg.build_graph(s["positions"], s["numbers"], s.get("cell"), s.get("pbc"), radius, max_neighbors,
                           dictionary)
        g["z"] = np.asarray(s["numbers"], dtype=np.int64)
        graphs.append(g)
    return from_graphs(graphs, ys, ids, num_edge_features)


def _synthetic(sizes, boxes, rng, seed, radius, max_neighbors, num_edge_features, tag):
    """Shared body of the synthetic generators: uniform positions in an orthorhombic periodic box per graph,
    minimum-image distances, the reference graph rule (process.py:284-305), Z ~ U[1,89], y ~ Normal(0,1) seed+1."""
    graphs = []
    for n, box in zip(sizes, boxes):
        box = np.asarray(box, dtype=np.float64)
        pos = rng.uniform(0.0, 1.0, size=(n, 3)) * box[:3]
        if len(box) > 3:                                     # slab: atoms only in the lowest box[3] of the z axis
            pos[:, 2] *= box[3] / box[2]
        numbers = rng.integers(1, 90, size=n)
        d = pos[None, :, :] - pos[:, None, :]
        d -= box[:3] * np.rint(d / box[:3])                  # minimum image, orthorhombic cell
        dm = np.sqrt((d * d).sum(-1))
        ei, ew = pg.edges_from_trimmed(pg.threshold_sort(dm, radius, max_neighbors))
        x = np.concatenate([pg.atom_features(numbers), pg.one_hot_degree(ei, n, max_neighbors + 1)], 1)
        graphs.append({"x": x, "edge_index": ei, "edge_weight": ew, "z": numbers})
    ys = np.random.default_rng(seed + 1).normal(0.0, 1.0, size=(len(sizes), 1)).astype(np.float32)
    return from_graphs(graphs, ys, ["%s%d" % (tag, i) for i in range(len(sizes))], num_edge_features)


def synthetic_bulk(n_graphs=46744, seed=0, density=0.05, mean_atoms=20.0, sigma=0.7, min_atoms=1, max_atoms=200,
                   radius=8.0, max_neighbors=12, num_edge_features=50):
    """Synthetic "bulk-like" stand-in for the absent Materials-Project bulk_data:
    n ~ clip(round(exp(Normal(ln mean_atoms, sigma))), min, max); cubic periodic cell of side
    (n/density)^(1/3); uniform positions; the reference graph rule; Z ~ U[1,89]; y ~ Normal(0,1) seed+1."""
    rng = np.random.default_rng(seed)
    sizes = np.clip(np.rint(np.exp(rng.normal(np.log(mean_atoms), sigma, n_graphs))), min_atoms, max_atoms).astype(int)
    boxes = [((n / density) ** (1.0 / 3.0),) * 3 for n in sizes]
    return _synthetic(sizes, boxes, rng, seed, radius, max_neighbors, num_edge_features, "syn")


def synthetic_mof(n_graphs=18000, seed=0, density=0.03, mean_atoms=100.0, sigma=0.5, min_atoms=20, max_atoms=500,
                  radius=8.0, max_neighbors=12, num_edge_features=50):
    """Synthetic "MOF-like" stand-in for the absent MOF_data: 18 k porous graphs of 20-500 atoms,
    n ~ clip(round(exp(Normal(ln 100, 0.5))), 20, 500), density 0.03 atoms/A^3, the reference graph rule."""
    rng = np.random.default_rng(seed)
    sizes = np.clip(np.rint(np.exp(rng.normal(np.log(mean_atoms), sigma, n_graphs))), min_atoms, max_atoms).astype(int)
    boxes = [((n / density) ** (1.0 / 3.0),) * 3 for n in sizes]
    return _synthetic(sizes, boxes, rng, seed, radius, max_neighbors, num_edge_features, "mof")


def synthetic_surface(n_graphs=37000, seed=0, density=0.06, min_atoms=40, max_atoms=80, slab=8.0, vacuum=15.0,
                      radius=8.0, max_neighbors=12, num_edge_features=50):
    """Synthetic "surface-like" stand-in for the absent surface_data: periodic slabs of 40-80 atoms,
    `slab` A thick with `vacuum` A of empty space along z (the minimum image never crosses the vacuum)."""
    rng = np.random.default_rng(seed)
    sizes = rng.integers(min_atoms, max_atoms + 1, size=n_graphs)
    boxes = []
    for n in sizes:
        a = (n / (density * slab)) ** 0.5
        boxes.append((a, a, slab + vacuum, slab))
    return _synthetic(sizes, boxes, rng, seed, radius, max_neighbors, num_edge_features, "surf")


class StaticBatch:
    """Fixed-shape (padded) batch buffers for HIP-graph replays of a training step.

    A captured graph freezes every launch argument and every address, while batches differ in node / edge count.  So
    the batch lives in buffers sized for `n_cap` nodes / `e_cap` edges; `load(ids)` (host side, outside the graph) uploads
    the graph ids and the prefix offsets, `assemble()` (inside the graph) runs K8 + the tail padding + K1 into the same
    buffers.  Padding nodes have no edges and belong to a dummy graph `B` (the models see num_graphs = B + 1; the caller
    drops the last prediction); the number of rows that exist is `n_dev`, a device scalar that the row-count-dependent
    kernels (BatchNorm) read through ops.true_rows().  Edge slots past the batch's last edge are referenced by no rowptr
    range."""

    def __init__(self, ds, batch_size, n_cap, e_cap, x_dtype=torch.float32, edge_dtype=torch.float32, ring=8, by_source=True):
        """by_source=False: the model never walks the edges by source (CGCNN: its backward reaches the source rows with atomics), so
        assemble() skips the by-source CSR of the batch — one launch per step less; a model that does ask for it then fails
        loudly on a stale index instead of reading one."""
        if ds.device is None or ds.device.type != "cuda":
            raise ops.MdlError("StaticBatch: the dataset must be resident on a HIP device")
        self.by_source = bool(by_source)
        dev = ds.device
        B, F, G = int(batch_size), ds.num_features, ds.num_edge_features
        self.ds, self.B, self.n_cap, self.e_cap = ds, B, int(n_cap), int(e_cap)
        # ids [B] | noff [B+1] | eoff [B+1] (int64) | the pooling index's row pointers [B+2] as int32 (two per word): one upload
        self._npack = 3 * B + 2
        self.pack_all = torch.zeros(self._npack + (B + 3) // 2, dtype=torch.int64, device=dev)
        self.pack = self.pack_all[:self._npack]
        self.n_dev = self.pack[2 * B:2 * B + 1]                                # noff[B] = number of nodes that exist
        self.e_dev = self.pack[3 * B + 1:3 * B + 2]
        self.x = torch.zeros((self.n_cap, F), dtype=x_dtype, device=dev)
        self.batch_idx = torch.full((self.n_cap,), B, dtype=torch.int64, device=dev)
        self.rowptr = torch.zeros(self.n_cap + 1, dtype=torch.int32, device=dev)
        self.src = torch.zeros(self.e_cap, dtype=torch.int32, device=dev)
        self.tgt = torch.zeros(self.e_cap, dtype=torch.int32, device=dev)
        self.ew = torch.zeros(self.e_cap, dtype=torch.float32, device=dev)
        self.dn = torch.zeros(self.e_cap, dtype=torch.float32, device=dev)
        self.y = torch.zeros(B, dtype=torch.float32, device=dev)
        self.edge_attr = torch.zeros((self.e_cap, G), dtype=edge_dtype, device=dev)
        csr = ops.EdgeCSR(self.rowptr, self.src, self.tgt, None, self.n_cap, self.e_cap)
        # by-source index (static buffers, filled inside the graph) and the int64 edge_index view some models read
        self.rowptr_s = torch.zeros(self.n_cap + 1, dtype=torch.int32, device=dev)
        self.col_s, self.eid_s, self.src_s = (torch.zeros(self.e_cap, dtype=torch.int32, device=dev) for _ in range(3))
        if self.by_source:
            csr.set_transposed((self.rowptr_s, self.col_s, self.eid_s, self.src_s))
        else:
            def _no_by_source():
                raise ops.MdlError("this StaticBatch was built with by_source=False (the model declared needs_by_source = False): "
                                   "the by-source CSR of the batch is not maintained")
            csr.set_transposed_builder(_no_by_source)
        csr.partial = True                                     # rows past n_dev / e_dev belong to no segment
        self.edge_index = torch.zeros((2, self.e_cap), dtype=torch.int64, device=dev)
        self.b_dev = torch.full((1,), B, dtype=torch.int64, device=dev)
        ds.by_source()
        # BatchNorm finds the true row count of a tensor by its PADDED row count (ops.true_rows): the three capacities must
        # be pairwise distinct, or node-, edge- and graph-level tensors would be normalised over each other's row counts
        if len({self.n_cap, self.e_cap, B + 1}) != 3:
            raise ops.MdlError("StaticBatch: node capacity %d, edge capacity %d and graph rows %d must be pairwise distinct"
                               % (self.n_cap, self.e_cap, B + 1))
        self.pool_rowptr = self.pack_all[self._npack:].view(torch.int32)[:B + 2]    # (uploaded with the ids: no launch)
        self.pool_seg = torch.full((self.n_cap,), B, dtype=torch.int32, device=dev)
        self.batch = Batch(pool_index=ops.make_seg_index(self.pool_rowptr, self.pool_seg, partial=True), x=self.x,
                           true_rows={self.n_cap: self.n_dev, self.e_cap: self.e_dev, B + 1: self.b_dev}, edge_attr=self.edge_attr, edge_weight=self.ew, batch=self.batch_idx, y=self.y,
                           u=torch.zeros(B + 1, 3, device=dev), num_graphs=B + 1, csr=csr, num_nodes=self.n_cap,
                           num_edges=self.e_cap, n_dev=self.n_dev, structure_id=None)
        self.batch._edge_index = self.edge_index

        def _fill_edge_index():
            self.edge_index[0].copy_(self.src)
            self.edge_index[1].copy_(self.tgt)
        self.batch._edge_index_fill = _fill_edge_index
        self.batch._edge_index_stale = True
        ops.register_csr(self.edge_index, csr)
        # every index tensor a model may hand to scatter() maps to the loader's segment index: no sorts inside the graph,
        # and the unused tail of the buffers stays outside every segment
        if self.by_source:
            ops.register_seg_index(self.edge_index[0], csr.seg_src(), owner=self.edge_index)
            ops.register_seg_index(self.src, csr.seg_src())
        ops.register_seg_index(self.edge_index[1], csr.seg_tgt(), owner=self.edge_index)
        ops.register_seg_index(self.tgt, csr.seg_tgt())
        ops.register_seg_index(self.batch_idx, self.batch.pool_index)
        self._pinned = [torch.zeros(self.pack_all.numel(), dtype=torch.int64).pin_memory() for _ in range(ring)]
        self._events = [None] * ring
        self._slot = 0
        self.true_nodes = self.true_edges = 0

    def fits(self, ids):
        ids = np.asarray(ids, dtype=np.int64)
        n = int((self.ds.node_ptr[ids + 1] - self.ds.node_ptr[ids]).sum())
        e = int((self.ds.edge_ptr[ids + 1] - self.ds.edge_ptr[ids]).sum())
        return len(ids) == self.B and n < self.n_cap and e <= self.e_cap       # (one padding node always exists)

    def load(self, ids):
        """Host side of a step: ids + exclusive prefix sums -> the device `pack` (one async copy from a pinned ring)."""
        ds = self.ds
        ids = np.asarray(ids, dtype=np.int64)
        noff = np.concatenate([[0], np.cumsum(ds.node_ptr[ids + 1] - ds.node_ptr[ids])])
        eoff = np.concatenate([[0], np.cumsum(ds.edge_ptr[ids + 1] - ds.edge_ptr[ids])])
        if len(ids) != self.B or noff[-1] >= self.n_cap or eoff[-1] > self.e_cap:
            raise ops.MdlError("StaticBatch.load: batch (%d graphs, %d nodes, %d edges) exceeds the static capacity "
                               "(%d, %d, %d)" % (len(ids), noff[-1], eoff[-1], self.B, self.n_cap, self.e_cap))
        k = self._slot
        self._slot = (k + 1) % len(self._pinned)
        if self._events[k] is not None:
            self._events[k].synchronize()                       # the copy that last used this pinned slot has finished
        # node -> graph pooling index straight from the prefix offsets.  The dummy graph B is EMPTY (the padding rows belong to
        # no segment): as one segment of thousands of rows it would be walked by a single lane group
        prp = np.zeros(2 * ((self.B + 3) // 2), dtype=np.int32)
        prp[:self.B + 1] = noff
        prp[self.B + 1] = noff[-1]
        self._pinned[k].copy_(torch.from_numpy(np.concatenate([ids, noff, eoff, prp.view(np.int64)])))
        self.pack_all.copy_(self._pinned[k], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._events[k] = ev
        self.true_nodes, self.true_edges = int(noff[-1]), int(eoff[-1])
        # every load changes src / tgt on the next assembly (eager or replayed): the lazily filled int64 edge_index of the static
        # batch is stale from here on, whether or not the Python body of assemble() runs again (it does not on replays)
        self.batch._edge_index_stale = True

    def assemble(self):
        """Device side (part of the captured graph): K8 batch assembly, tail padding, K1 RBF expansion."""
        from .. import _lib
        ds, B = self.ds, self.B
        d = ds._dev
        p = _lib.ptr
        ids_d, noff_d, eoff_d = self.pack[:B], self.pack[B:2 * B + 1], self.pack[2 * B + 1:]
        # K8 with the tails of the padded buffers closed by extra workgroups of the same launch (padding nodes: no edges, dummy graph
        # B; unused edge slots -> the first padding node) and the pooling index's int32 segment ids written beside `batch`: one
        # launch where round 5 had four (mdl_assemble_batch, mdl_pad_batch_tail, mdl_pad_edge_tail, an int64 -> int32 copy)
        bs = self.by_source
        _lib.check(_lib.lib().mdl_assemble_batch_padded(
            p(ids_d), p(noff_d), p(eoff_d), p(d["node_ptr"]), p(d["edge_ptr"]), p(d["x"]), p(d["src"]), p(d["tgt"]),
            p(d["dist"]), p(d["dist_norm"]), p(d["lrowptr"]), p(d["y"]), p(self.x), p(self.batch_idx), p(self.rowptr),
            p(self.src), p(self.tgt), p(self.ew), p(self.dn), p(self.y), B, ds.num_features, ds.y.shape[1],
            int(ds.target_index), _lib.dtype_code(self.x), self.n_cap, self.e_cap, p(self.pool_seg),
            p(self.col_s) if bs else None, p(self.eid_s) if bs else None, p(self.src_s) if bs else None, _lib.stream()),
            "mdl_assemble_batch_padded")
        if bs:
            eperm_s, lrowptr_s = ds.by_source()
            _lib.check(_lib.lib().mdl_assemble_transposed(
                p(ids_d), p(noff_d), p(eoff_d), p(d["node_ptr"]), p(d["edge_ptr"]), p(d["src"]), p(d["tgt"]), p(eperm_s),
                p(lrowptr_s), p(self.rowptr_s), p(self.col_s), p(self.eid_s), p(self.src_s), B, self.n_cap, _lib.stream()),
                "mdl_assemble_transposed")
        self.batch._edge_index_stale = True       # the int64 [2, e_cap] view: refilled when (and if) a model reads batch.edge_index
        ops.rbf_expand(self.dn, 0.0, 1.0, ds.num_edge_features, 0.2, offsets=d["offsets"], out=self.edge_attr)
        # the static CSR outlives the batch: its work-balance prefix (CGConv backward) is rebuilt with the batch, in place
        self.batch.csr.refresh_balance()
        return self.batch


def static_capacity(ds, batch_size, indices=None, slack=6.0, quantum=None):
    """(n_cap, e_cap) for StaticBatch: mean + `slack` standard deviations of a random batch's node / edge count, rounded
    up — a batch that still exceeds it takes the eager path.  quantum None: 1024 rows for large capacities, 256 / 64 for
    small ones (at the reference's batch size a 1024-row quantum alone is 40 % of padding: every padded tile is a tile of
    work for the conv kernels)."""
    idx = np.arange(len(ds)) if indices is None else np.asarray(indices)
    nn = (ds.node_ptr[idx + 1] - ds.node_ptr[idx]).astype(np.float64)
    ne = (ds.edge_ptr[idx + 1] - ds.edge_ptr[idx]).astype(np.float64)
    B = float(batch_size)

    def cap(v):
        raw = B * v.mean() + slack * v.std() * np.sqrt(B)
        q = quantum if quantum else (1024 if raw >= 32768 else 256 if raw >= 4096 else 64)
        return int(-(-raw // q) * q)
    return cap(nn), cap(ne)


class DeviceLoader:
    """Mini-batch iterator over a subset of a device-resident GraphDataset.

    Shuffling reseeds from (seed, epoch) like torch's DistributedSampler (same contract, numpy's bit generator); with world_size > 1 rank r
    takes perm[r::world_size] of the (padded) index list — the DistributedSampler contract the
    reference relies on at training.py:291-294 — and every rank draws `batch_size` graphs per step."""

    def __init__(self, dataset, indices, batch_size, shuffle=False, seed=0, rank=0, world_size=1,
                 edge_dtype=torch.float32, rbf=None, prefetch=None):
        self.ds, self.indices = dataset, np.asarray(indices, dtype=np.int64)
        self.batch_size, self.shuffle, self.seed = int(batch_size), shuffle, int(seed)
        self.rank, self.world_size, self.edge_dtype, self.rbf = rank, world_size, edge_dtype, rbf
        self.epoch = 0
        # assemble batch k + 1 on a side stream while the caller works on batch k (HIP datasets with the kernel RBF expansion).
        # Ownership: a yielded batch belongs to the stream that was current when the iterator handed it out (take_ahead orders
        # that stream behind the assembly and records it with the allocator); a consumer on another stream calls
        # Batch.record_stream(stream) — see there.
        on_hip = dataset.device is not None and dataset.device.type == "cuda"
        self.prefetch = (on_hip and rbf is None) if prefetch is None else (bool(prefetch) and on_hip and rbf is None)

    def set_epoch(self, epoch):
        self.epoch = int(epoch)

    def _order(self):
        idx = self.indices
        if self.shuffle:
            # numpy's generator, not torch.randperm: torch's CPU ops enter the intra-op thread pool, and on a many-core
            # host shared with other jobs waking 128 threads for a 37 k-element permutation stalled the training thread
            # for milliseconds — every epoch boundary, in the middle of an otherwise device-bound step stream
            idx = idx[np.random.default_rng(self.seed + self.epoch).permutation(len(idx))]
        if self.world_size > 1:
            total = -(-len(idx) // self.world_size) * self.world_size
            idx = np.concatenate([idx, idx[: total - len(idx)]])[self.rank::self.world_size]
        return idx

    def __len__(self):
        n = len(self.indices) if self.world_size == 1 else -(-len(self.indices) // self.world_size)
        return -(-n // self.batch_size)

    def batch_ids(self):
        """The epoch's batches as graph-id arrays, in the order __iter__ assembles them (training.GraphedStep takes ids: it
        assembles the batch inside its captured step)."""
        idx = self._order()
        for i in range(0, len(idx), self.batch_size):
            yield idx[i:i + self.batch_size]

    def __iter__(self):
        idx = self._order()
        starts = list(range(0, len(idx), self.batch_size))
        if not self.prefetch or torch.cuda.is_current_stream_capturing():
            for i in starts:
                yield self.ds.collate(idx[i:i + self.batch_size], self.edge_dtype, self.rbf, x_dtype=self.edge_dtype)
            return
        ahead = None
        for k, i in enumerate(starts):
            if ahead is None:
                ahead = self.ds.collate_ahead(idx[i:i + self.batch_size], self.edge_dtype, x_dtype=self.edge_dtype)
            batch = self.ds.take_ahead(ahead)
            ahead = None
            if k + 1 < len(starts):
                j = starts[k + 1]
                ahead = self.ds.collate_ahead(idx[j:j + self.batch_size], self.edge_dtype, x_dtype=self.edge_dtype)
            yield batch
