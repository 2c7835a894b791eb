"""Multi-GPU substeps: one process and one libmpmhip context per GPU (not in the reference,
which is single-GPU).

Decomposition
  * ownership: vertices and traditional particles are assigned to ranks by the quantile slab of their x coordinate at
    the last (re-)partition, an element to the owner of its first vertex.  Between re-partitions ownership is fixed (the
    ghost lists below are built once per partition; particles that wander into a neighbour's slab only widen the halo).
    MIGRATION: ownership is a matter of performance, not of correctness -- any two ranks exchange whatever blocks they share -- and a
    body that only MOVES (a garment walking across the slabs) keeps its halo.  ``maybe_repartition`` (called at the start of every
    ``run``) therefore asks first whether the halo has GROWN since the partition was cut (``migrate_halo_factor``, one all-reduce) and
    only then whether more than ``migrate_fraction`` of the owned particles have left their slab; if both, it gathers the state of all
    ranks, cuts new slabs at the current positions and rebuilds every rank's shard -- a stop-the-world step that costs about as much
    as the initial build and is needed only when material mixes across a cut (poured sand).
  * ghosts: a rank also holds (a) every element that touches one of its vertices (so vertex forces are complete
    without an exchange) and (b) every vertex of its local elements.  Ghost copies carry particle_selection == 2:
    stress / element finalise run on them, p2g / g2p do not.
  * per substep ONE neighbour exchange: after p2g the (m, momentum[, mover]) channels of the grid blocks that both
    ranks have on their active lists are summed; both ranks then evaluate the identical grid stage there.  Ghost
    copies gather for themselves (g2p yes, p2g no): their grid neighbourhood is on both active lists, hence complete
    after the sum.  Owner and copy differ only by the rounding order of those sums; at every collective re-sort the
    owners overwrite the copies (x, v of vertices, d3 of elements) to keep that from accumulating.
    (MPMHIP_DIST_GHOST_G2P=0 selects the older scheme: copies do not gather, owners send them every substep.)
    The body-face splat is replicated (each rank splats the faces that touch its active blocks).
  * all ranks re-sort at the same substep: every ``rebin_interval`` substeps if that is > 0, otherwise (default) when the
    max over the ranks of the library's early-warning drift flag asks for it (polled every 16 substeps, at the latest
    every 256) -- the single-GPU adaptive policy made collective; the shared-block lists are rebuilt
    there from an all_gather of the per-rank active-block maps.
Transports: "rccl" (default with the nccl backend) runs the whole substep loop inside libmpmhip.so with its own
RCCL communicator (ncclSend/ncclRecv groups on the solver stream, ncclAllGather of the block maps; no Python per
substep); "torch" drives the three phases from Python with torch.distributed P2P ops (MPMHIP_DIST_TRANSPORT=torch).
Backend "nccl" (= RCCL) exchanges device buffers; "gloo" stages through host memory and exists so that the whole
path can be exercised with two processes on one GPU (tests/test_gpu_dist.py) and on CPU-only CI (partition logic).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field, replace
from typing import Dict, List

import numpy as np

from .scenes import Scene


@dataclass
class Shard:
    rank: int
    world: int
    scene: Scene                      # local scene: owned + ghost particles, joint entries first
    own_e: np.ndarray                 # global element ids owned (ascending)
    ghost_e: np.ndarray               # global element ids held as ghosts
    own_t: np.ndarray
    own_v: np.ndarray
    ghost_v: np.ndarray
    send_p: Dict[int, np.ndarray] = field(default_factory=dict)   # peer -> local particle indices (vertices) to send
    recv_p: Dict[int, np.ndarray] = field(default_factory=dict)
    send_e: Dict[int, np.ndarray] = field(default_factory=dict)   # peer -> local element indices whose d3 is sent
    recv_e: Dict[int, np.ndarray] = field(default_factory=dict)
    send_p_gid: Dict[int, np.ndarray] = field(default_factory=dict)  # the same lists as global ids (tests)
    recv_p_gid: Dict[int, np.ndarray] = field(default_factory=dict)
    cuts: np.ndarray = None           # the world-1 slab boundaries (x) of this partition
    t_gid: np.ndarray = None          # local traditional row -> global traditional id, -1 = free row (slack for on-device migration);
                                      # free rows first, owned rows behind them in ascending id order (own_t = t_gid[t_gid >= 0])


_TEST_FAIL_BUILD_RANK = None   # test hook (tests/dist_worker.py sets it): the rank whose local build raises; never read from the environment


# Free rows a rank's traditional class is built with (fraction of its owned particles, at least TRAD_SLACK_MIN): what on-device
# migration moves particles INTO.  Free rows carry particle_selection = 1 (never simulated, sorted behind everything, no chunk).
TRAD_SLACK, TRAD_SLACK_MIN = 0.25, 256


CUT_BINS = 1 << 14   # histogram resolution of the device-side quantile cut: grid_lim / 16384 (1/64 cell at 256^3)


def cuts_from_histogram(hist, world: int, grid_lim: float) -> np.ndarray:
    """The world-1 slab boundaries from a histogram of the owned particles' x (CUT_BINS bins over [0, grid_lim), summed over the
    ranks): the upper edge of the first bin at which the cumulative count reaches k/world of the particles.  Deterministic and
    identical on every rank (integer counts); within one bin width of np.quantile."""
    h = np.asarray(hist, np.int64)
    cum = np.cumsum(h)
    total = int(cum[-1]) if cum.size else 0
    if world <= 1 or total == 0:
        return np.zeros(0)
    targets = (np.arange(1, world) * total + world - 1) // world
    idx = np.searchsorted(cum, targets, side="left")
    return (idx + 1).astype(np.float64) * (float(grid_lim) / h.size)


def device_cuts(ss: "ShardedSim") -> np.ndarray:
    """Collective: new slab boundaries at the particles' CURRENT positions without bringing a position to the host: every rank
    histograms the x of the particles it owns on its GPU (torch.histc), the histograms are summed over the ranks (one all-reduce of
    64 KiB) and every rank reads the same cuts off the same cumulative counts.  (Round 3: np.quantile over the all-gathered
    positions of the whole scene.)"""
    import torch
    import torch.distributed as dist
    sh = ss.shard
    xs = _owned_movable_x(ss).float()
    lim = float(ss.global_scene.grid_lim)
    h = torch.histc(xs, bins=CUT_BINS, min=0.0, max=lim) if xs.numel() else torch.zeros(CUT_BINS, device=x.device)
    h = h.to(torch.int64)
    h = h.cpu() if ss.backend == "gloo" else h
    dist.all_reduce(h)
    return cuts_from_histogram(h.cpu().numpy(), sh.world, lim)


def _owned_movable_x(ss: "ShardedSim"):
    """x of the traditional particles and vertices this rank OWNS (device tensor): traditional rows by their global-id table (free
    rows of the migration slack excluded), then the owned vertices (the first own_v rows of the vertex class)."""
    import torch
    sh = ss.shard
    x = ss.sim.state.particle_x.detach()[:, 0]
    ne_l = sh.own_e.size + sh.ghost_e.size
    nt_l = sh.t_gid.size
    own_rows = torch.as_tensor(np.nonzero(sh.t_gid >= 0)[0], device=x.device, dtype=torch.int64)
    xt = x[ne_l:ne_l + nt_l][own_rows] if nt_l else x[:0]
    return torch.cat([xt, x[ne_l + nt_l:ne_l + nt_l + sh.own_v.size]])


def _owners(sc: Scene, world: int, cuts=None):
    n_e, n_t = sc.n_elements, sc.n_traditional
    xt, xv = sc.x[n_e:n_e + n_t, 0], sc.x[n_e + n_t:, 0]
    px = np.concatenate([xv, xt]).astype(np.float64)
    if cuts is None:
        cuts = np.quantile(px, np.arange(1, world) / world) if world > 1 and px.size else np.zeros(0)
    owner_v = np.searchsorted(cuts, xv, side="right").astype(np.int32)
    owner_t = np.searchsorted(cuts, xt, side="right").astype(np.int32)
    owner_e = owner_v[sc.faces[:, 0]] if n_e else np.zeros(0, np.int32)
    return owner_e, owner_t, owner_v, np.asarray(cuts, np.float64)


def partition(sc: Scene, world: int, cuts=None, slack: bool = True) -> List[Shard]:
    """Deterministic: every rank computes the full partition from the same Scene, no communication.  cuts: slab boundaries to use
    (re-partition: device_cuts()); None = the world-quantiles of the vertices' and traditional particles' x.  slack: scenes with
    traditional particles get free rows in every rank's traditional class (TRAD_SLACK) for on-device migration."""
    n_e, n_t, n_v = sc.n_elements, sc.n_traditional, sc.n_vertices
    owner_e, owner_t, owner_v, cuts = _owners(sc, world, cuts)
    faces = sc.faces.astype(np.int64)
    shards = []
    for r in range(world):
        own_e = np.nonzero(owner_e == r)[0]
        touches = (owner_v[faces] == r).any(1) if n_e else np.zeros(0, bool)
        ghost_e = np.nonzero(touches & (owner_e != r))[0]
        el = np.concatenate([own_e, ghost_e])
        own_v = np.nonzero(owner_v == r)[0]
        need_v = np.unique(faces[el].reshape(-1)) if el.size else np.zeros(0, np.int64)
        ghost_v = need_v[owner_v[need_v] != r]
        vl = np.concatenate([own_v, ghost_v])
        own_t = np.nonzero(owner_t == r)[0]
        n_free = (int(TRAD_SLACK * own_t.size) + TRAD_SLACK_MIN) if (slack and world > 1 and n_t > 0 and sc.joint_t_hold >= 0) else 0
        # local traditional rows: free rows first (a copy of some valid particle, never simulated), the owned ones behind them in
        # ascending id order -- the mover's "last n traditional particles" (run_demo.py:524) stays a suffix of the owned ones
        t_src = np.concatenate([np.full(n_free, own_t[0] if own_t.size else 0, np.int64), own_t])
        t_gid = np.concatenate([np.full(n_free, -1, np.int64), own_t.astype(np.int64)])
        g2l = np.full(n_v, -1, np.int64)
        g2l[vl] = np.arange(vl.size)
        f_loc = g2l[faces[el]].astype(np.int32) if el.size else np.zeros((0, 3), np.int32)
        assert (f_loc >= 0).all()
        x = np.concatenate([sc.x[el], sc.x[n_e + t_src], sc.x[n_e + n_t + vl]], 0)
        v = np.concatenate([sc.v[el], sc.v[n_e + t_src], sc.v[n_e + n_t + vl]], 0)
        vol = np.concatenate([sc.vol[el], sc.vol[n_e + t_src], sc.vol[n_e + n_t + vl]], 0)
        sel = np.zeros(x.shape[0], np.int32)
        sel[own_e.size:el.size] = 2
        sel[el.size:el.size + n_free] = 1
        sel[el.size + t_src.size + own_v.size:] = 2
        njv = int((own_v < sc.num_joint_v).sum())
        njf = int((own_e < sc.num_joint_f).sum())
        jv = None if sc.joint_verts_v is None else sc.joint_verts_v[own_v[:njv]]
        jf = None if sc.joint_faces_v is None else sc.joint_faces_v[own_e[:njf]]
        local = replace(sc, name=f"{sc.name}[{r}/{world}]", n_elements=int(el.size), n_traditional=int(t_src.size),
                        n_vertices=int(vl.size), x=np.ascontiguousarray(x, np.float32), v=np.ascontiguousarray(v, np.float32),
                        vol=np.ascontiguousarray(vol, np.float32), faces=f_loc, d=sc.d[el], R_inv=sc.R_inv[el],
                        num_joint_v=njv, num_joint_f=njf, joint_verts_v=jv, joint_faces_v=jf, selection=sel, joint_t_hold=0,
                        has_mover=(sc.num_joint_v > 0 or sc.num_joint_f > 0) if sc.has_mover is None else sc.has_mover)
        shards.append(Shard(r, world, local, own_e, ghost_e, own_t, own_v, ghost_v, cuts=cuts, t_gid=t_gid))
    # ghost exchange lists, ordered by global id on both sides
    for r, sh in enumerate(shards):
        off_v = sh.scene.n_elements + sh.scene.n_traditional   # (traditional class incl. its free rows)
        lv = {g: i for i, g in enumerate(np.concatenate([sh.own_v, sh.ghost_v]))}
        le = {g: i for i, g in enumerate(np.concatenate([sh.own_e, sh.ghost_e]))}
        for q, other in enumerate(shards):
            if q == r:
                continue
            sv = np.intersect1d(other.ghost_v, sh.own_v)           # I own, q holds as ghost
            rv = np.intersect1d(sh.ghost_v, other.own_v)           # q owns, I hold as ghost
            se = np.intersect1d(other.ghost_e, sh.own_e)
            re_ = np.intersect1d(sh.ghost_e, other.own_e)
            if sv.size + rv.size + se.size + re_.size == 0:
                continue
            sh.send_p[q] = np.array([off_v + lv[g] for g in sv], np.int32)
            sh.recv_p[q] = np.array([off_v + lv[g] for g in rv], np.int32)
            sh.send_e[q] = np.array([le[g] for g in se], np.int32)
            sh.recv_e[q] = np.array([le[g] for g in re_], np.int32)
            sh.send_p_gid[q], sh.recv_p_gid[q] = sv, rv
    return shards


# --------------------------------------------------------------------------------------------- runtime
@dataclass
class ShardedSim:
    shard: Shard
    sim: object                        # harness.Sim of the local scene
    backend: str
    rebin_interval: int
    transport: str = "torch"           # "rccl": loop inside libmpmhip.so; "torch": phases driven from Python
    ghost_g2p: bool = True             # ghost copies gather for themselves (one exchange per substep)
    steps_done: int = 0
    since: int = 0                     # substeps since the last collective re-sort
    resort_now: bool = False           # some rank's drift flag was up at the last poll
    sorted_once: bool = False
    resorts: int = 0
    peers: list = field(default_factory=list)
    keep: list = field(default_factory=list)
    static: dict = field(default_factory=dict)
    global_scene: Scene = None         # the unsharded scene this shard was cut from (positions as of the last partition)
    migrate_fraction: float = 0.10     # re-partition when more than this fraction of the particles left their slab (0: never) ...
    migrate_halo_factor: float = 1.5   # ... AND the halo -- the bytes a rank sends per substep -- has grown by this factor since the
                                       # partition was cut (0: do not ask).  A body that MOVES keeps its halo: a garment walking across
                                       # the slabs, a cube thrown along x leave "their" slabs entirely without a single block more to
                                       # exchange, and ownership is a matter of performance only (any two ranks exchange whatever blocks
                                       # they share); only material that MIXES across the cut (poured sand) makes the halo grow
    halo_ref: int = 0                  # max over the ranks of the halo bytes per substep right after the partition's first re-sort
    migrate_check_every: int = 512     # ... looked at every this many substeps (one device-side count + one all-reduce)
    migrate_checked_at: int = 0
    migrate_warned: bool = False
    migrations: int = 0
    migrate_trad_fraction: float = 0.01   # on-device migration of traditional particles when at least this fraction of them lies in
                                          # another rank's slab of the CURRENT quantile cuts (0: at every check; < 0: never)
    trad_migrations: int = 0              # on-device migration events / particles moved by them / wall time spent in them
    trad_migrated: int = 0
    trad_migration_ms: list = field(default_factory=list)   # wall time of every event (the first pays torch's one-time kernel loads)
    mass_version: tuple = ()           # (data_ptr, version counter) of state.particle_mass when the ranks last agreed on the scene's mass span


def sync_mass_span(ss: "ShardedSim") -> float:
    """Collective: the smallest positive and the largest mass over the SIMULATED particles of all ranks, from the mass tensors as they
    are bound now (after reset_density(update_mass=True), after a re-partition's carried masses), handed to every rank's context
    (mpmhip_dist_set_mass_span).  MPMHIP_P2G_TILE_AUTO then switches to the fp64 tile on every rank together when the span of the
    WHOLE scene exceeds 1e5 -- a rank deciding from its own shard could run other accumulator numerics than its neighbour on the halo
    blocks they share, and a decision taken from the scene description would miss masses changed after the build.
    Returns the span."""
    import torch
    import torch.distributed as dist
    st, sv = ss.sim.state, ss.sim.solver
    mt = st._raw("particle_mass")
    m = mt.detach().float().reshape(-1)
    ok = m > 0
    sel = st._raw("particle_selection")
    if sel is not None and sel.numel() == m.numel():
        ok = ok & (sel.reshape(-1) == 0)
    big = torch.finfo(torch.float32).max
    lo = torch.where(ok, m, torch.full_like(m, big)).min() if m.numel() else torch.tensor(big, device=m.device)
    hi = torch.where(ok, m, torch.zeros_like(m)).max() if m.numel() else torch.tensor(0.0, device=m.device)
    t = torch.stack([-lo, hi]).to(torch.float32)      # one MAX all-reduce: max(-lo) = -min(lo)
    t = t.cpu() if ss.backend == "gloo" else t
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    lo_g, hi_g = -float(t[0].item()), float(t[1].item())
    if hi_g <= 0.0 or lo_g >= big:
        lo_g, hi_g = 0.0, 0.0                           # no simulated particle anywhere
    sv._call("mpmhip_dist_set_mass_span", float(lo_g), float(hi_g))
    ss.mass_version = (mt.data_ptr(), mt._version)
    return hi_g / lo_g if lo_g > 0 else 1.0


def build_sharded(sc: Scene, device, rank: int, world: int, rebin_interval: int = 0, _carry: dict = None, _cuts=None) -> ShardedSim:
    """Collective.  ``_carry`` / ``_cuts`` (re-partition only): per-particle state in GLOBAL order to continue from and the slab
    boundaries to cut at, see repartition()."""
    import os
    import torch
    import torch.distributed as dist
    from . import harness
    dev = torch.device(device)
    backend = dist.get_backend()
    # (1) Everything that can fail on ONE rank alone -- host memory for the partition, device memory for the solver and the exchange
    # buffers -- runs BEFORE the first collective of the build, and the ranks vote on it: a rank that raised here used to go straight
    # to its caller's all_reduce while the others were inside the peer-handle exchange below (mismatched collectives: a hang or
    # undefined results on RCCL instead of the clean collective error repartition() promises).
    ss, err = None, None
    try:
        if _TEST_FAIL_BUILD_RANK == rank:   # (set by tests/dist_worker.py only: a rank whose local build fails)
            raise MemoryError("injected failure of the local build (tests)")
        shard = partition(sc, world, _cuts)[rank]
        # MPMHIP_P2G_TILE_AUTO means "the SCENE's masses span more than 1e5" (include/mpmhip.h).  The context stays on AUTO; the span it
        # decides from is made global right after the collective part below (sync_mass_span) and again whenever the masses change.
        sim = harness.build_solver(shard.scene, device, mode="fast")
        sv = sim.solver
        if _carry is not None:
            _apply_carry(sim, shard, sc, _carry)
        sv._bind(sim.model, sim.state)
        sv._call("mpmhip_dist_enable")
        ghost_g2p = os.environ.get("MPMHIP_DIST_GHOST_G2P", "1") != "0"
        sv._call("mpmhip_dist_set_ghost_mode", 1 if ghost_g2p else 0)
        ss = ShardedSim(shard, sim, backend, int(rebin_interval), ghost_g2p=ghost_g2p)
        ss.global_scene = sc
        i32 = lambda a: torch.as_tensor(np.ascontiguousarray(a, np.int32), device=dev)
        for q in sorted(set(shard.send_p) | set(shard.recv_p)):
            ss.static[q] = dict(send_p=i32(shard.send_p[q]), recv_p=i32(shard.recv_p[q]), send_e=i32(shard.send_e[q]),
                                recv_e=i32(shard.recv_e[q]))
            n_s = 6 * len(shard.send_p[q]) + 3 * len(shard.send_e[q])
            n_r = 6 * len(shard.recv_p[q]) + 3 * len(shard.recv_e[q])
            ss.static[q]["gs"] = torch.zeros(max(n_s, 1), dtype=torch.float32, device=dev)
            ss.static[q]["gr"] = torch.zeros(max(n_r, 1), dtype=torch.float32, device=dev)
        nb = sv._lib.mpmhip_dist_num_blocks(sv._ctx)
        ss.static["map"] = torch.zeros(nb, dtype=torch.uint8, device=dev)
    except Exception as e:  # noqa: BLE001 - reported collectively right below
        err = e
    if world > 1:
        ok = torch.tensor([0 if err is not None else 1], dtype=torch.int32)
        if backend == "nccl":
            ok = ok.to(dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) != 1:
            if ss is not None:
                ss.sim.solver.close()
            raise RuntimeError("mpmavatar_amd.dist.build_sharded: the local part of the build failed on "
                               + ("this rank: " + repr(err) if err is not None else "another rank") + " (every rank raises)") from err
    elif err is not None:
        raise err
    # (2) the collective part: transport set-up (communicator, peer links)
    want = os.environ.get("MPMHIP_DIST_TRANSPORT", "rccl" if ss.backend == "nccl" else "torch")
    if want == "rccl":
        # every rank must end up on the same transport: agree on success before switching
        ok = 1
        try:
            _init_rccl(ss, rank, world)
        except Exception as e:  # noqa: BLE001 - any failure means "use the torch transport"
            print(f"[mpmavatar_amd.dist] rank {rank}: in-library RCCL transport unavailable ({e}); using torch.distributed", flush=True)
            ok = 0
        flag = torch.tensor([ok], dtype=torch.int32, device=dev if ss.backend == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        ss.transport = "rccl" if int(flag.item()) == 1 else "torch"
    sync_mass_span(ss)
    return ss


# --------------------------------------------------------------------------------------------- migration
# per-particle state a re-partition carries over (particle_mass: reset_density(update_mass=True) after the initial build;
# mu / lam / yield_stress below: set_E_nu* / prepare_mu_lam after the build and the plastic history).  NOT carried: edits of
# particle_selection made after build_sharded, and the host-side state of grid BCs (a moving velocity cuboid's position) --
# scenes with such a BC are not re-partitioned (maybe_repartition warns once and goes on).
_CARRY_FIELDS = ("particle_x", "particle_v", "particle_C", "particle_d", "particle_F", "particle_F_trial", "particle_stress",
                 "particle_mass")
_CARRY_MODEL = ("mu", "lam", "yield_stress")


def local_to_global_rows(shard: Shard, sc: Scene):
    """Global particle index of every OWNED local particle, and the local rows they sit in: (rows_local, ids_global) for the
    all-particle arrays (x, v, C, model arrays) and for the element+traditional ones (F, F_trial, stress)."""
    ne_g, nt_g = sc.n_elements, sc.n_traditional
    ne_l = shard.own_e.size + shard.ghost_e.size
    nt_l = shard.t_gid.size                                  # local traditional rows incl. free ones
    t_loc = np.nonzero(shard.t_gid >= 0)[0]                  # ... the owned ones among them
    rows_e = np.arange(shard.own_e.size)
    rows_t = ne_l + t_loc
    rows_v = ne_l + nt_l + np.arange(shard.own_v.size)
    all_rows = np.concatenate([rows_e, rows_t, rows_v])
    all_ids = np.concatenate([shard.own_e, ne_g + shard.t_gid[t_loc], ne_g + nt_g + shard.own_v])
    nv_rows = np.concatenate([rows_e, rows_t])
    nv_ids = np.concatenate([shard.own_e, ne_g + shard.t_gid[t_loc]])
    return all_rows, all_ids, nv_rows, nv_ids


def owned_slices(local: dict, shard: Shard, sc: Scene) -> dict:
    """The rows of a rank's local arrays (local particle order) that belong to particles it OWNS, with their global ids."""
    all_rows, all_ids, nv_rows, nv_ids = local_to_global_rows(shard, sc)
    n_loc = shard.scene.n_particles
    out = {"all_ids": all_ids, "nv_ids": nv_ids, "e_ids": shard.own_e}
    for f, a in local.items():
        if f == "particle_d":
            out[f] = a[:shard.own_e.size]
        elif f in _CARRY_MODEL or f in ("particle_x", "particle_v", "particle_C", "particle_mass"):
            assert a.shape[0] == n_loc, (f, a.shape, n_loc)
            out[f] = a[all_rows]
        else:
            out[f] = a[nv_rows]
    return out


def assemble_global(parts, sc: Scene) -> dict:
    """Per-rank owned slices -> arrays in the global (unsharded) particle order."""
    n_p, n_nv, n_e = sc.n_particles, sc.n_elements + sc.n_traditional, sc.n_elements
    out = {}
    for f in _CARRY_FIELDS + _CARRY_MODEL:
        if f not in parts[0]:
            continue
        whole = f in _CARRY_MODEL or f in ("particle_x", "particle_v", "particle_C", "particle_mass")
        n = n_p if whole else (n_e if f == "particle_d" else n_nv)
        g = np.zeros((n,) + parts[0][f].shape[1:], np.float32)
        filled = np.zeros(n, bool)
        for p in parts:
            ids = p["all_ids"] if whole else (p["e_ids"] if f == "particle_d" else p["nv_ids"])
            g[ids] = p[f]
            filled[ids] = True
        assert filled.all(), f"{f}: {int((~filled).sum())} particles have no owner"
        out[f] = g
    return out


def gather_global_state(ss: "ShardedSim") -> dict:
    """Collective: the current state of every particle, assembled from its owner, in the global (unsharded) order."""
    import torch.distributed as dist
    st, md, sh, sc = ss.sim.state, ss.sim.model, ss.shard, ss.global_scene
    local = {f: getattr(st, f).detach().cpu().numpy() for f in _CARRY_FIELDS}
    local.update({f: getattr(md, f).detach().cpu().numpy() for f in _CARRY_MODEL})
    parts = [None] * sh.world
    dist.all_gather_object(parts, owned_slices(local, sh, sc))
    return assemble_global(parts, sc)


def _apply_carry(sim, shard: Shard, sc: Scene, carry: dict):
    """Write the carried state of this rank's (owned and ghost) particles into the freshly built local solver."""
    import torch
    st, md = sim.state, sim.model
    ne_g, nt_g = sc.n_elements, sc.n_traditional
    el = np.concatenate([shard.own_e, shard.ghost_e])
    vl = np.concatenate([shard.own_v, shard.ghost_v])
    t_ids = np.where(shard.t_gid >= 0, shard.t_gid, shard.own_t[0] if shard.own_t.size else 0)   # (free rows: any valid particle's state)
    all_ids = np.concatenate([el, ne_g + t_ids, ne_g + nt_g + vl])
    nv_ids = np.concatenate([el, ne_g + t_ids])
    dev = st.particle_x.device
    put = lambda dst, a: dst.copy_(torch.as_tensor(np.ascontiguousarray(a, np.float32), device=dev).reshape(dst.shape)) if dst.numel() else None
    for f in ("particle_C", "particle_mass"):
        put(getattr(st, f), carry[f][all_ids])
    for f in ("particle_F", "particle_F_trial", "particle_stress"):
        put(getattr(st, f), carry[f][nv_ids])
    for f in _CARRY_MODEL:
        put(getattr(md, f), carry[f][all_ids])
    sim.solver.time = carry["time"]


def slab_leavers(ss: "ShardedSim") -> float:
    """Collective: fraction of all vertices / traditional particles that sit outside the x-slab of the rank that owns them.
    Counted on the device (no copy of the positions to the host); one two-element all-reduce."""
    import torch
    import torch.distributed as dist
    sh = ss.shard
    xs = _owned_movable_x(ss)
    n_own = sh.own_t.size + sh.own_v.size
    lo = -np.inf if sh.rank == 0 else float(sh.cuts[sh.rank - 1])
    hi = np.inf if sh.rank == sh.world - 1 else float(sh.cuts[sh.rank])
    out = ((xs < lo) | (xs >= hi)).sum().to(torch.float64)
    # (collectives run at world size 1 too: a tensor on the wrong device for the backend -- NCCL takes no CPU tensors --
    # then fails on a one-GPU test box and not first on the node)
    t = torch.stack([out, torch.tensor(float(n_own), dtype=torch.float64, device=out.device)])
    t = t.cpu() if ss.backend == "gloo" else t
    dist.all_reduce(t)
    t = t.cpu()
    return float(t[0] / max(float(t[1]), 1.0))


def halo_bytes_max(ss: "ShardedSim") -> int:
    """Collective: max over the ranks of the bytes a rank sends per substep in the halo exchange (shared blocks x channels x 256 B),
    as of the last collective re-sort."""
    import torch
    import torch.distributed as dist
    if ss.transport == "rccl":
        hb = C.c_int64(0)
        ss.sim.solver._call("mpmhip_dist_halo_bytes", C.byref(hb))
        mine = int(hb.value)
    else:
        mine = 4 * sum(int(p["n_halo"]) for p in ss.peers)
    t = torch.tensor([mine], dtype=torch.int64, device="cpu" if ss.backend == "gloo" else ss.sim.solver.device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return int(t.item())


# --------------------------------------------------------------------------------------------- on-device migration (traditional particles)
_MIG_STATE = (("particle_x", 3), ("particle_v", 3), ("particle_C", 9), ("particle_F", 9), ("particle_F_trial", 9), ("particle_stress", 9),
              ("particle_vol", 1), ("particle_mass", 1), ("particle_density", 1))
_MIG_MODEL = ("E", "nu", "mu", "lam", "gamma", "kappa", "yield_stress")
MIG_RECORD_FLOATS = sum(w for _, w in _MIG_STATE) + len(_MIG_MODEL)   # 52 floats = 208 B per migrating particle (+ its 8-byte global id)


def _mig_tensors(ss: "ShardedSim"):
    """The per-particle float tensors a traditional particle's record is made of, each as a [local traditional rows, width] VIEW of the
    caller-order tensor the solver is bound to."""
    st, md, sh = ss.sim.state, ss.sim.model, ss.shard
    ne_l, nt_l = sh.own_e.size + sh.ghost_e.size, sh.t_gid.size
    out = []
    for name, w in _MIG_STATE:
        out.append(st._raw(name).detach().reshape(-1, w)[ne_l:ne_l + nt_l])
    for name in _MIG_MODEL:
        out.append(md._raw(name).detach().reshape(-1, 1)[ne_l:ne_l + nt_l])
    return out


def migrate_traditional(ss: "ShardedSim", min_fraction: float = 0.0) -> int:
    """Collective, ON THE DEVICE: new slab boundaries at the current positions (device_cuts: quantiles of a histogram summed over the
    ranks, so a body that only moves takes its cuts with it and nothing migrates), and every owned traditional particle that now lies in
    another rank's slab goes there -- its record (52 floats + global id) through ONE all-to-all of device tensors, into a free row of the
    destination's traditional class (TRAD_SLACK) -- and leaves a free row behind.  No state goes through the host, no context is
    rebuilt; what the host learns is one histogram, a world x world table of counts and the new id table of its own rows.  The
    traditional block is left sorted (free rows first, owned rows by ascending global id): the mover's "last n particles" rule
    (run_demo.py:524) keeps holding per rank.  Elements and vertices are not touched (cloth ownership follows the mesh, not the slab;
    see maybe_repartition).
    Returns the number of particles that changed rank (all ranks: the same number); 0 when fewer than min_fraction of the traditional
    particles would move; -1 when some rank has too few free rows (nothing is changed then: the caller falls back to repartition())."""
    import torch
    import torch.distributed as dist
    sh, sim = ss.shard, ss.sim
    sc = ss.global_scene
    world, rank = sh.world, sh.rank
    if world == 1 or sc.n_traditional == 0 or sh.t_gid is None:
        return 0
    sv, st = sim.solver, sim.state
    dev = st._raw("particle_x").device
    host = ss.backend == "gloo"
    import os
    import time
    marks = []

    def mark(what):   # MPMHIP_VERBOSE=2: where a migration event's time goes (synchronises the device at every mark)
        if os.environ.get("MPMHIP_VERBOSE") == "2":
            torch.cuda.synchronize()
            marks.append((what, time.perf_counter()))
    mark("start")
    sv._call("mpmhip_pull_state")                       # the caller-order tensors are the solver's state now (one export kernel)
    mark("pull")
    cuts_np = device_cuts(ss)                           # (collective) quantile cuts of the owned vertices + traditional particles
    mark("cuts")
    cuts = torch.as_tensor(cuts_np, dtype=torch.float32, device=dev)
    gid = torch.as_tensor(sh.t_gid, dtype=torch.int64, device=dev)
    owned = gid >= 0
    fields = _mig_tensors(ss)
    x = fields[0][:, 0]
    dest = torch.bucketize(x.contiguous(), cuts, right=True)       # = np.searchsorted(cuts, x, side="right"), the rule of _owners()
    move = owned & (dest != rank)
    out_counts = torch.bincount(dest[move], minlength=world).to(torch.int64)
    table = [torch.zeros(world, dtype=torch.int64, device="cpu" if host else dev) for _ in range(world)]
    dist.all_gather(table, out_counts.cpu() if host else out_counts)
    table = torch.stack(table).cpu().numpy()            # table[src][dst]
    n_moving = int(table.sum())
    n_in = int(table[:, rank].sum())
    n_free = int((~owned).sum().item())
    ok = torch.tensor([1 if n_in <= n_free + int(table[rank].sum()) else 0], dtype=torch.int32, device="cpu" if host else dev)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    mark("counts")
    if n_moving == 0 or n_moving < min_fraction * sc.n_traditional:
        return 0
    if int(ok.item()) != 1:
        return -1
    # records of the leavers, grouped by destination (stable: ascending row inside a group)
    idx = torch.nonzero(move).reshape(-1)
    idx = idx[torch.sort(dest[idx], stable=True).indices]
    rec = torch.cat([f[idx] for f in fields], dim=1).contiguous()
    rec_id = gid[idx].contiguous()
    out_split = [int(c) for c in table[rank]]
    in_split = [int(c) for c in table[:, rank]]
    rec_in = torch.empty((n_in, MIG_RECORD_FLOATS), dtype=torch.float32, device=dev)
    id_in = torch.empty(n_in, dtype=torch.int64, device=dev)
    if host:   # gloo: staged through host memory like every other exchange of the test backend
        ro, io = list(rec.cpu().split(out_split)), list(rec_id.cpu().split(out_split))
        ri, ii = [torch.empty((c, MIG_RECORD_FLOATS)) for c in in_split], [torch.empty(c, dtype=torch.int64) for c in in_split]
        ops = []   # (gloo has no all-to-all: point-to-point, batched)
        for q in range(world):
            if q == rank:
                continue
            for src, dst in ((ro, ri), (io, ii)):
                if src[q].numel():
                    ops.append(dist.P2POp(dist.isend, src[q].contiguous(), q))
                if dst[q].numel():
                    ops.append(dist.P2POp(dist.irecv, dst[q], q))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        rec_in, id_in = torch.cat(ri).to(dev), torch.cat(ii).to(dev)
    else:
        dist.all_to_all_single(rec_in, rec, in_split, out_split)
        dist.all_to_all_single(id_in, rec_id, in_split, out_split)
    mark("exchange")
    # leavers free their rows, arrivals take free rows (the leavers' included)
    gid[idx] = -1
    free_rows = torch.nonzero(gid < 0).reshape(-1)[:n_in]
    off = 0
    for f in fields:
        w = f.shape[1]
        f[free_rows] = rec_in[:, off:off + w]
        off += w
    gid[free_rows] = id_in
    mark("insert")
    # sorted layout: free rows first, owned rows by ascending global id
    order = torch.sort(gid, stable=True).indices
    for f in fields:
        f.copy_(f[order])
    gid = gid[order]
    ne_l = sh.own_e.size + sh.ghost_e.size
    sel = st._raw("particle_selection")
    sel[ne_l:ne_l + gid.numel()] = (gid < 0).to(sel.dtype)      # 1: free row (never simulated), 0: owned
    mark("sort rows")
    sh.t_gid = gid.cpu().numpy()
    sh.own_t = sh.t_gid[sh.t_gid >= 0]
    sh.cuts = np.asarray(cuts_np, np.float64)
    st._touch()
    sim.model._touch()
    sv._call("mpmhip_push_state")                       # the caller-order tensors are newer than the solver's sorted copy: re-import at the
    ss.sorted_once = False                              # collective re-sort that the next substep starts with
    ss.mass_version = ()                                # (masses moved with their particles: the ranks agree on the span again)
    ss.trad_migrated += n_moving
    ss.trad_migrations += 1
    mark("bookkeeping")
    if marks and rank == 0:
        print("[mpmavatar_amd.dist] migration event: " + ", ".join(f"{b[0]} {1e3 * (b[1] - a[1]):.2f} ms" for a, b in zip(marks, marks[1:])), flush=True)
    return n_moving


def repartition(ss: "ShardedSim") -> "ShardedSim":
    """Collective: new slabs at the particles' CURRENT positions; every rank rebuilds its shard and continues the same run
    (state, solver time and substep count carried over).  The ShardedSim is updated IN PLACE (and returned): a caller that
    keeps its reference without rebinding continues on the new shard.  The old solver context -- with its RCCL communicator
    and peer-mapped arenas -- is destroyed before the new shard is built, so the two never coexist."""
    import sys
    import torch
    import torch.distributed as dist
    sc = ss.global_scene
    cuts = device_cuts(ss)                      # (before the state leaves the device: histogram of the owned particles' x)
    carry = gather_global_state(ss)
    carry["time"] = ss.sim.solver.time
    new_sc = replace(sc, x=carry["particle_x"], v=carry["particle_v"], d=carry["particle_d"])
    dev = str(ss.sim.solver.device)
    keep = dict(steps_done=ss.steps_done, resorts=ss.resorts, migrate_fraction=ss.migrate_fraction, migrations=ss.migrations + 1,
                migrate_halo_factor=ss.migrate_halo_factor, halo_ref=0, migrate_trad_fraction=ss.migrate_trad_fraction,
                trad_migrations=ss.trad_migrations, trad_migrated=ss.trad_migrated, trad_migration_ms=ss.trad_migration_ms,
                migrate_check_every=ss.migrate_check_every, migrate_checked_at=ss.migrate_checked_at)
    rank, world, rebin_interval = ss.shard.rank, ss.shard.world, ss.rebin_interval
    old = ss.sim
    ss.sim, ss.peers, ss.keep, ss.static = None, [], [], {}     # drop every tensor the old context was handed ...
    for name in ("keep_body", "keep_jt", "keep_body_frame"):
        ss.__dict__.pop(name, None)
    old.solver.close()                                          # ... and the context itself (communicator, IPC arenas)
    del old
    # The rebuild can fail on ONE rank (out of memory, a peer link that does not come up): that rank must not be left with
    # sim = None while the others wait in the next collective.  Every rank reports, the minimum decides, all raise.
    new, err = None, None
    try:
        new = build_sharded(new_sc, dev, rank, world, rebin_interval=rebin_interval, _carry=carry, _cuts=cuts)
    except Exception as e:  # noqa: BLE001 - reported collectively below
        err = e
    ok = torch.tensor([0 if new is None else 1], dtype=torch.int32, device="cpu" if ss.backend == "gloo" else dev)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if int(ok.item()) != 1:
        if new is not None:
            new.sim.solver.close()
        msg = f"re-partition at substep {keep['steps_done']} failed on " + ("this rank: " + repr(err) if err is not None else "another rank")
        print(f"[mpmavatar_amd.dist] rank {rank}: {msg}", file=sys.stderr, flush=True)
        raise RuntimeError("mpmavatar_amd.dist.repartition: " + msg + " (every rank raises; the sharded simulation is gone -- "
                           "rebuild it from a checkpoint of gather_global_state)") from err
    ss.__dict__.update(new.__dict__)
    for k, v in keep.items():
        setattr(ss, k, v)
    ss.sim.steps_done = keep["steps_done"]
    return ss


def maybe_repartition(ss: "ShardedSim") -> "ShardedSim":
    import os
    import sys
    if (ss.migrate_fraction <= 0 and ss.migrate_trad_fraction < 0) or ss.shard.world == 1 or ss.steps_done == 0:
        return ss
    if ss.steps_done - ss.migrate_checked_at < ss.migrate_check_every:
        return ss
    ss.migrate_checked_at = ss.steps_done
    if any(kind == "velocity_cuboid" for kind, _ in ss.global_scene.bcs):
        # the cuboid's host-side position (set_velocity_on_cuboid's modify step) is not part of what a re-partition carries over
        if not ss.migrate_warned and ss.shard.rank == 0:
            print("[mpmavatar_amd.dist] scene has a moving velocity cuboid: particle migration is skipped (slabs stay as cut at "
                  "the start)", file=sys.stderr, flush=True)
        ss.migrate_warned = True
        return ss
    # (0) traditional particles that have mixed across the cuts change rank ON THE DEVICE (migrate_traditional): cheap enough to do at
    # every look, and what it does not fix (too few free rows; cloth folded across a cut) is left to the halo criterion below
    if ss.migrate_trad_fraction >= 0 and ss.global_scene.n_traditional > 0:
        import time
        import torch
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        moved = migrate_traditional(ss, ss.migrate_trad_fraction)
        torch.cuda.synchronize()
        if moved > 0:
            ss.trad_migration_ms.append(1e3 * (time.perf_counter() - t0))
        if os.environ.get("MPMHIP_VERBOSE"):
            print(f"[mpmavatar_amd.dist] rank {ss.shard.rank}: substep {ss.steps_done}: on-device migration of traditional particles: "
                  + ("too few free rows on some rank" if moved < 0 else f"{moved} changed rank"), flush=True)
        if moved > 0:
            ss.halo_ref = 0          # (new slabs: the halo reference is taken again at the next look)
            return ss
    if ss.migrate_fraction <= 0:
        return ss
    # (1) has the halo grown?  One int64 all-reduce; the reference value is taken at the first look after a (re-)partition
    halo = halo_bytes_max(ss)
    if ss.halo_ref <= 0:
        ss.halo_ref = max(halo, 1)
    grown = ss.migrate_halo_factor <= 0 or halo > ss.migrate_halo_factor * ss.halo_ref
    if not grown:
        if os.environ.get("MPMHIP_VERBOSE"):
            print(f"[mpmavatar_amd.dist] rank {ss.shard.rank}: substep {ss.steps_done}: halo {halo} B per substep (x{halo / ss.halo_ref:.2f} of "
                  f"the partition's {ss.halo_ref} B): slabs stay", flush=True)
        return ss
    # (2) ... and is it because particles left their slabs (new cuts at the current positions would help)?
    frac = slab_leavers(ss)
    if os.environ.get("MPMHIP_VERBOSE"):
        print(f"[mpmavatar_amd.dist] rank {ss.shard.rank}: substep {ss.steps_done}: halo x{halo / ss.halo_ref:.2f}, {100 * frac:.1f} % of the "
              f"particles are outside their owner's slab" + (" -> re-partition" if frac > ss.migrate_fraction else ""), flush=True)
    return repartition(ss) if frac > ss.migrate_fraction else ss


def _init_rccl(ss: ShardedSim, rank: int, world: int):
    """Create the library's own RCCL communicator (unique id from rank 0, broadcast through torch.distributed) and hand
    it the static ghost lists."""
    import torch.distributed as dist
    from . import _lib as L
    sv, sh = ss.sim.solver, ss.shard
    uid = (C.c_char * 128)()
    if rank == 0:
        L.check(sv._lib, None, sv._lib.mpmhip_rccl_unique_id(uid))
    box = [bytes(uid.raw) if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    uid = (C.c_char * 128).from_buffer_copy(box[0])
    sv._call("mpmhip_rccl_init", rank, world, uid)
    peers = sorted(set(sh.send_p) | set(sh.recv_p))
    n = len(peers)
    I = lambda vals: (C.c_int32 * max(n, 1))(*vals)
    keep = []

    def ptrs(d):
        arr = (L.ip * max(n, 1))()
        for i, q in enumerate(peers):
            a = np.ascontiguousarray(d[q], np.int32)
            keep.append(a)
            arr[i] = a.ctypes.data_as(L.ip)
        return arr
    sv._call("mpmhip_rccl_set_ghosts", n, I(peers), I([len(sh.send_p[q]) for q in peers]), ptrs(sh.send_p),
             I([len(sh.recv_p[q]) for q in peers]), ptrs(sh.recv_p), I([len(sh.send_e[q]) for q in peers]), ptrs(sh.send_e),
             I([len(sh.recv_e[q]) for q in peers]), ptrs(sh.recv_e))
    ss.transport = "rccl"


def _all_gather_maps(ss: ShardedSim):
    import torch
    import torch.distributed as dist
    m = ss.static["map"]
    world = ss.shard.world
    if ss.backend == "gloo":
        mc = m.cpu()
        outs = [torch.zeros_like(mc) for _ in range(world)]
        dist.all_gather(outs, mc)
        return [o.to(m.device) for o in outs]
    outs = [torch.zeros_like(m) for _ in range(world)]
    dist.all_gather(outs, m)
    return outs


def rebin_all(ss: ShardedSim):
    """Collective: every rank re-sorts and rebuilds the shared-block lists with each peer."""
    import torch
    from . import _lib as L
    sv = ss.sim.solver
    sv._call("mpmhip_dist_rebin", ss.static["map"].data_ptr())
    maps = _all_gather_maps(ss)
    mine = maps[ss.shard.rank]
    ch = 8 if sv.particle_movers else 4
    peers, keep = [], []
    arr = []
    for q in range(ss.shard.world):
        if q == ss.shard.rank:
            continue
        blocks = torch.nonzero(mine & maps[q]).reshape(-1).to(torch.int32)
        st = ss.static.get(q)
        if blocks.numel() == 0 and st is None:
            continue
        nb = int(blocks.numel())
        hs = torch.zeros(max(nb * ch * 64, 1), dtype=torch.float32, device=mine.device)
        hr = torch.zeros_like(hs)
        keep += [blocks, hs, hr]
        p = L.DistPeer()
        p.n_blocks, p.blocks, p.halo_send, p.halo_recv = nb, blocks.data_ptr() if nb else None, hs.data_ptr(), hr.data_ptr()
        if st is not None:
            p.n_send_p, p.n_recv_p = st["send_p"].numel(), st["recv_p"].numel()
            p.n_send_e, p.n_recv_e = st["send_e"].numel(), st["recv_e"].numel()
            p.send_p, p.recv_p = st["send_p"].data_ptr(), st["recv_p"].data_ptr()
            p.send_e, p.recv_e = st["send_e"].data_ptr(), st["recv_e"].data_ptr()
            p.ghost_send, p.ghost_recv = st["gs"].data_ptr(), st["gr"].data_ptr()
        arr.append(p)
        peers.append(dict(rank=q, hs=hs, hr=hr, n_halo=nb * ch * 64,
                          gs=None if st is None else st["gs"], gr=None if st is None else st["gr"],
                          n_gs=0 if st is None else 6 * st["send_p"].numel() + 3 * st["send_e"].numel(),
                          n_gr=0 if st is None else 6 * st["recv_p"].numel() + 3 * st["recv_e"].numel()))
    carr = (L.DistPeer * max(len(arr), 1))(*arr)
    sv._call("mpmhip_dist_set_peers", len(arr), carr)
    ss.peers, ss.keep = peers, keep


def _exchange(ss: ShardedSim, kind: str):
    import torch
    import torch.distributed as dist
    ops, post = [], []
    for p in ss.peers:
        sbuf, rbuf, ns, nr = (p["hs"], p["hr"], p["n_halo"], p["n_halo"]) if kind == "halo" else (p["gs"], p["gr"], p["n_gs"], p["n_gr"])
        if ss.backend == "gloo":
            if ns:
                ops.append(dist.P2POp(dist.isend, sbuf[:ns].cpu(), p["rank"]))
            if nr:
                tmp = torch.empty(nr, dtype=torch.float32)
                ops.append(dist.P2POp(dist.irecv, tmp, p["rank"]))
                post.append((rbuf, tmp, nr))
        else:
            if ns:
                ops.append(dist.P2POp(dist.isend, sbuf[:ns], p["rank"]))
            if nr:
                ops.append(dist.P2POp(dist.irecv, rbuf[:nr], p["rank"]))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    for rbuf, tmp, nr in post:
        rbuf[:nr].copy_(tmp)


def _any_rank_drifting(ss: ShardedSim) -> bool:
    """Collective: max over the ranks of the library's early-warning drift flag."""
    import torch
    import torch.distributed as dist
    sv = ss.sim.solver
    flag = C.c_int32(0)
    sv._call("mpmhip_dist_drift_flag", C.byref(flag))
    t = torch.tensor([flag.value], dtype=torch.int32, device="cpu" if ss.backend == "gloo" else sv.device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return bool(int(t.item()))


def _held_local(ss: ShardedSim, step: int) -> int:
    """This rank's share of the traditional particles the mover still holds at substep `step` (run_demo.py:524: the last
    joint_t_count(step) traditional particles in the global order; owned ids are ascending, so the share is a suffix)."""
    sc, sh = ss.global_scene, ss.shard
    if sc.joint_t_hold <= 0:
        return -1
    return int((sh.own_t >= sc.n_traditional - sc.joint_t_count(step)).sum())


def run(ss: ShardedSim, n_steps: int):
    # NOTE: the migration check (maybe_repartition) runs at the START of a call only, every `migrate_check_every` (512) substeps of
    # accumulated progress: run(ss, n) with a large n never re-partitions mid-call -- callers that expect particles to cross slabs
    # advance in frames (the reference's drivers call per frame of 400 substeps anyway).
    """Advance n substeps on every rank (collective).  Returns ``ss`` (a re-partition updates it in place)."""
    import torch
    ss = maybe_repartition(ss)
    sim, sv, sc = ss.sim, ss.sim.solver, ss.sim.scene
    mt = sim.state._raw("particle_mass")
    if (mt.data_ptr(), mt._version) != ss.mass_version:   # masses rewritten since the ranks last agreed (reset_density(update_mass))
        sync_mass_span(ss)                                    # (collective: every rank rewrites its masses through the same caller code)
    dp = lambda t: None if t is None or t.numel() == 0 else t.data_ptr()
    jv, jf = sim.joint_verts_v, sim.joint_faces_v
    dummy = sv._dummy_ptr()
    if sv._host_dt != sc.dt:  # MPMWARP.time advances by the Python float (mpm_solver.py:536)
        sv._call("mpmhip_set_host_dt", float(sc.dt))
        sv._host_dt = sc.dt
    held = lambda step: _held_local(ss, step)
    gsc = ss.global_scene
    swaying = getattr(gsc, "mesh_sway", None) is not None and sim.mesh_x0 is not None
    t3 = lambda a: torch.as_tensor(np.ascontiguousarray(a, np.float32), device=sv.device).reshape(-1, 3)

    def body(step):
        """(mesh_x at substep 0 of the library's advection x + (step dt) v, mesh_v, joint_verts_v, joint_faces_v) valid at
        `step`.  A body posed per frame (Scene.mesh_sway, train_material_params.py:617-622) has a new velocity every frame:
        the frame's start pose is extrapolated back to substep 0 with that velocity, so that the same absolute advection
        factor (step_index + k) dt reproduces pose(frame start) + (step - frame start) dt v."""
        if not swaying:
            return sim.mesh_x0, sim.mesh_v, jv, jf
        f0, _ = gsc.frame_of(step)
        if getattr(ss, "keep_body_frame", None) == f0:
            return ss.keep_body
        ss.keep_body_frame = f0
        mx, mv = gsc.body_at(f0)
        x0 = (mx.astype(np.float64) - gsc.dt * f0 * mv.astype(np.float64)).astype(np.float32)
        bjv, bjf = jv, jf
        if jv is not None:
            vel = torch.as_tensor(np.ascontiguousarray(mv[0], np.float32), device=sv.device)
            bjv, bjf = vel.expand(jv.shape[0], 3).contiguous(), vel.expand(jf.reshape(-1, 3).shape[0], 3).contiguous()
        ss.keep_body = (t3(x0), t3(mv), bjv, bjf)   # alive until the next frame's tensors replace them
        return ss.keep_body

    def frame_run(step, n):
        """Substeps from `step` that share one body pose / velocity."""
        if not swaying:
            return n
        f0, spf = gsc.frame_of(step)
        return min(n, f0 + spf - step)
    jt_buf = None
    if ss.global_scene.joint_t_hold > 0:
        jt_buf = torch.zeros((max(ss.shard.t_gid.size, 1), 3), dtype=torch.float32, device=sv.device)
        ss.keep_jt = jt_buf
    if ss.transport == "rccl":
        jvp = None if jv is None else (dp(jv) or dummy)
        jfp = None if jf is None else (dp(jf) or dummy)
        k = 0
        while k < n_steps:  # runs of substeps over which the number of held particles does not change
            n = n_steps - k
            h = held(ss.steps_done)
            if h >= 0:
                n = next((j for j in range(1, n) if held(ss.steps_done + j) != h), n)
            n = frame_run(ss.steps_done, n)
            bx, bv, bjv, bjf = body(ss.steps_done)
            if swaying:
                jvp = None if bjv is None else (dp(bjv) or dummy)
                jfp = None if bjf is None else (dp(bjf) or dummy)
            sv._call("mpmhip_rccl_steps", float(sc.dt), int(n), int(ss.steps_done), int(ss.rebin_interval),
                     dp(bx), dp(bv), jt_buf.data_ptr() if h > 0 else None, max(h, 0), jvp, jfp)
            ss.steps_done += n
            k += n
        return ss
    adaptive = ss.rebin_interval <= 0   # re-sort when any rank's drift flag asks for it (polled every 16 substeps)
    cap = -ss.rebin_interval if ss.rebin_interval < 0 else (256 if ss.rebin_interval == 0 else ss.rebin_interval)
    for _ in range(n_steps):
        due = (ss.resort_now or ss.since >= cap) if adaptive else ss.steps_done % cap == 0
        due = due or not ss.sorted_once   # a fresh shard (initial build or re-partition) has no particle order yet
        if due:
            if ss.ghost_g2p and ss.steps_done > 0 and ss.peers:  # owners -> copies before the re-sort
                sv._call("mpmhip_dist_ghost_pack")
                _exchange(ss, "ghost")
                sv._call("mpmhip_dist_ghost_unpack")
            rebin_all(ss)
            ss.since, ss.resort_now, ss.sorted_once = 0, False, True
            ss.resorts += 1
        adv = float(np.float32(sc.dt * ss.steps_done))
        bx, bv, bjv, bjf = body(ss.steps_done)
        jvp = None if bjv is None else (dp(bjv) or dummy)
        jfp = None if bjf is None else (dp(bjf) or dummy)
        h = held(ss.steps_done)
        sv._call("mpmhip_dist_step_begin", float(sc.dt), dp(bx), dp(bv), adv,
                 jt_buf.data_ptr() if h > 0 else None, max(h, 0), jvp, jfp)
        _exchange(ss, "halo")
        sv._call("mpmhip_dist_step_mid")
        if not ss.ghost_g2p:
            _exchange(ss, "ghost")
        sv._call("mpmhip_dist_step_end")
        ss.steps_done += 1
        ss.since += 1
        if adaptive and ss.since % 16 == 0:
            ss.resort_now = _any_rank_drifting(ss)
    return ss


def gather_positions(ss: ShardedSim):
    """Owned particle positions with their global ids (for tests): (elements, traditional, vertices)."""
    st, sh, sc = ss.sim.state, ss.shard, ss.sim.scene
    x = st.particle_x.detach().cpu().numpy()
    ne, nt = sc.n_elements, sc.n_traditional
    t_loc = np.nonzero(sh.t_gid >= 0)[0]
    return dict(e_id=sh.own_e, e_x=x[:sh.own_e.size], t_id=sh.t_gid[t_loc], t_x=x[ne + t_loc],
                v_id=sh.own_v, v_x=x[ne + nt:ne + nt + sh.own_v.size])
