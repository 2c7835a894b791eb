"""Host logic of the particle migration (mpmavatar_amd/dist.py): owned slices of every rank's local arrays reassemble to the
global arrays, whatever the partition; a re-partition at moved positions yields slabs nobody has left."""
import numpy as np
import pytest

from mpmavatar_amd import dist as mdist
from mpmavatar_amd import scenes


def _global_fields(sc, rng):
    n_p, n_nv, n_e = sc.n_particles, sc.n_elements + sc.n_traditional, sc.n_elements
    return {"particle_x": sc.x.copy(), "particle_v": rng.standard_normal((n_p, 3)).astype(np.float32),
            "particle_C": rng.standard_normal((n_p, 3, 3)).astype(np.float32), "particle_d": rng.standard_normal((n_e, 3, 3)).astype(np.float32),
            "particle_F": rng.standard_normal((n_nv, 3, 3)).astype(np.float32), "particle_F_trial": rng.standard_normal((n_nv, 3, 3)).astype(np.float32),
            "particle_stress": rng.standard_normal((n_nv, 3, 3)).astype(np.float32), "mu": rng.random(n_p).astype(np.float32),
            "lam": rng.random(n_p).astype(np.float32), "yield_stress": rng.random(n_p).astype(np.float32)}


def _local_view(g, sh, sc):
    """What a rank's state tensors hold: owned + ghost particles in the local order el | trad | vert."""
    ne, nt = sc.n_elements, sc.n_traditional
    el = np.concatenate([sh.own_e, sh.ghost_e])
    vl = np.concatenate([sh.own_v, sh.ghost_v])
    t_ids = np.where(sh.t_gid >= 0, sh.t_gid, 0)       # (free rows of the migration slack hold some particle's values; never read back)
    all_ids = np.concatenate([el, ne + t_ids, ne + nt + vl])
    nv_ids = np.concatenate([el, ne + t_ids])
    out = {}
    for f, a in g.items():
        out[f] = a[el] if f == "particle_d" else (a[all_ids] if a.shape[0] == sc.n_particles else a[nv_ids])
    return out


@pytest.mark.parametrize("world", [2, 3, 5])
@pytest.mark.parametrize("scene", ["garment", "demo", "cube"])
def test_owned_slices_reassemble_to_the_global_state(world, scene):
    sc = {"garment": scenes.small_garment, "demo": lambda: scenes.demo_mix(n_grid=48, n_sheet=16, sand=(16, 3, 8)),
          "cube": scenes.small_cube}[scene]()
    g = _global_fields(sc, np.random.default_rng(5))
    shards = mdist.partition(sc, world)
    parts = [mdist.owned_slices(_local_view(g, sh, sc), sh, sc) for sh in shards]
    back = mdist.assemble_global(parts, sc)
    for f, a in g.items():
        assert np.array_equal(back[f], a), f
    # every particle has exactly one owner
    cnt = np.zeros(sc.n_particles, int)
    for p in parts:
        cnt[p["all_ids"]] += 1
    assert (cnt == 1).all()


def test_repartition_at_moved_positions_recentres_the_slabs():
    sc = scenes.small_cube()
    old = mdist.partition(sc, 2)
    moved = sc.x.copy()
    moved[:, 0] += 0.6 * (sc.x[:, 0].max() - sc.x[:, 0].min())      # the whole cube slides across the old cut
    n_left = sum(int((moved[sc.n_elements + sh.own_t, 0] >= sh.cuts[0]).sum()) if sh.rank == 0 else
                 int((moved[sc.n_elements + sh.own_t, 0] < sh.cuts[0]).sum()) for sh in old)
    assert n_left > 0.3 * sc.n_particles
    from dataclasses import replace
    new = mdist.partition(replace(sc, x=moved), 2)
    assert abs(new[0].own_t.size - new[1].own_t.size) <= 1                     # balanced again
    assert (moved[sc.n_elements + new[0].own_t, 0] < new[0].cuts[0]).all() and (moved[sc.n_elements + new[1].own_t, 0] >= new[1].cuts[0]).all()


def test_held_share_is_a_suffix_of_the_owned_particles():
    sc = scenes.demo_mix(n_grid=48, n_sheet=16, sand=(16, 3, 8), hold=(10, 5, 64))
    for world in (2, 3):
        shards = mdist.partition(sc, world)
        for step in (0, 12, 27, 200):
            held_global = set(range(sc.n_traditional - sc.joint_t_count(step), sc.n_traditional))
            tot = 0
            for sh in shards:
                ss = type("S", (), {"global_scene": sc, "shard": sh})()
                h = mdist._held_local(ss, step)
                assert set(sh.own_t[sh.own_t.size - h:].tolist()) == held_global & set(sh.own_t.tolist())
                tot += h
            assert tot == len(held_global)


def test_weak_scaling_workload_gives_every_rank_one_sheets_worth():
    """bench.py's weak-scaling scene (N stacked copies of the sheet, cut into N x-slabs): every rank owns 1/N of EVERY layer,
    i.e. one sheet's worth of particles, whatever N."""
    import numpy as np
    from mpmavatar_amd import dist as mdist, scenes
    for world in (2, 3):
        sc = scenes.sheet_stack(world, n=24, n_grid=32)
        one = scenes.sheet(n=24, n_grid=32)
        assert sc.n_particles == world * one.n_particles and sc.n_elements == world * one.n_elements
        shards = mdist.partition(sc, world)
        owned = [s.own_e.size + s.own_t.size + s.own_v.size for s in shards]
        assert sum(owned) == sc.n_particles
        assert max(owned) - min(owned) <= 0.1 * one.n_particles, owned          # balanced: about one sheet each
        nv1 = one.n_vertices
        for s in shards:                                                        # ... made of a strip of every layer
            layers = np.unique(s.own_v // nv1)
            assert layers.size == world, (world, layers)


def test_device_side_quantile_cuts_match_np_quantile_to_a_bin():
    """Round 4: the slab boundaries of a re-partition come from a histogram of the owned particles' x summed over the ranks
    (dist.device_cuts) instead of np.quantile over all-gathered positions: identical on every rank, within one bin of the quantile,
    and partition(cuts=...) splits the particles as evenly."""
    from mpmavatar_amd import dist as md
    sc = scenes.small_sheet()
    n_e, n_t = sc.n_elements, sc.n_traditional
    px = np.concatenate([sc.x[n_e + n_t:, 0], sc.x[n_e:n_e + n_t, 0]])
    for world in (2, 3, 4):
        # per-"rank" histograms of disjoint subsets, summed: what the all-reduce delivers
        parts = np.array_split(np.random.default_rng(world).permutation(px), world)
        h = sum(np.histogram(p, bins=md.CUT_BINS, range=(0.0, sc.grid_lim))[0] for p in parts)
        cuts = md.cuts_from_histogram(h, world, sc.grid_lim)
        q = np.quantile(px.astype(np.float64), np.arange(1, world) / world)
        assert cuts.shape == q.shape and np.abs(cuts - q).max() <= 1.5 * sc.grid_lim / md.CUT_BINS + np.diff(np.unique(px)).max()
        shards = md.partition(sc, world, cuts)
        sizes = [s.own_v.size + s.own_t.size for s in shards]
        assert sum(sizes) == px.size and max(sizes) - min(sizes) <= 0.1 * px.size / world + 48   # (lattice columns are 24 vertices)
        assert all(np.array_equal(s.cuts, cuts) for s in shards)
    assert md.cuts_from_histogram(np.zeros(md.CUT_BINS, np.int64), 4, 2.0).size == 0
