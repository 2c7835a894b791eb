"""float64 vectorised NumPy twin of the reference substep (TEST INFRASTRUCTURE ONLY).

Independent of ``mpm_oracle.c``: it uses convention-free closed forms (Gram-Schmidt QR with
q3 = q1 x q2, closed-form 2x2 polar rotation, LAPACK SVD) instead of restating Warp's qr3/svd3 +
sign flips, and float64 throughout.  Agreement between the two (tests/test_oracle_twin.py) was what
pinned the fp32 C oracle in round 1; since round 2 the reference's own kernel bodies, run over a NumPy stand-in of the warp module,
pin it (mpm_oracle.h, PINNING) and the twin is the second, convention-free witness.
It also generates the committed golden fixtures (tests/golden/make_golden.py).

Reference lines: stencil mpm_utils.py:499-526; p2g :484-557; grid update :561-572; g2p :716-857;
cloth :101-209; isotropic models :8-84, :362-399, :1017-1105; colliders mpm_solver.py:564-918.
"""
from __future__ import annotations

import math

import numpy as np

MATERIALS = {"jelly": 0, "metal": 1, "sand": 2, "foam": 3, "snow": 4, "plasticine": 5, "neo-hookean": 6, "cloth": 7}


def _normalize(a):
    n = np.linalg.norm(a, axis=-1, keepdims=True)
    return np.where(n > 0, a / np.where(n > 0, n, 1.0), 0.0)


def qr_gs(d):
    """Unique QR with R00,R11 >= 0 and det Q = +1 (what qr3 + the two sign flips produce)."""
    d1, d2, d3 = d[..., 0], d[..., 1], d[..., 2]
    q1 = d1 / np.linalg.norm(d1, axis=-1, keepdims=True)
    r01 = (q1 * d2).sum(-1)
    u2 = d2 - r01[..., None] * q1
    r11 = np.linalg.norm(u2, axis=-1)
    q2 = u2 / r11[..., None]
    q3 = np.cross(q1, q2)
    Q = np.stack([q1, q2, q3], -1)
    R = np.zeros_like(d)
    R[..., 0, 0] = np.linalg.norm(d1, axis=-1)
    R[..., 0, 1] = r01
    R[..., 1, 1] = r11
    R[..., 0, 2] = (q1 * d3).sum(-1)
    R[..., 1, 2] = (q2 * d3).sum(-1)
    R[..., 2, 2] = (q3 * d3).sum(-1)
    return Q, R


def stencil(x, inv_dx):
    gp = x * inv_dx
    base = np.trunc(gp - 0.5).astype(np.int64)
    fx = gp - base
    w = np.stack([0.5 * (1.5 - fx) ** 2, 0.75 - (fx - 1.0) ** 2, 0.5 * (fx - 0.5) ** 2], -1)  # [n, axis, node]
    dw = np.stack([fx - 1.5, -2.0 * (fx - 1.0), fx - 0.5], -1)
    return base, fx, w, dw


_IJK = np.stack(np.meshgrid(np.arange(3), np.arange(3), np.arange(3), indexing="ij"), -1).reshape(27, 3)


class TwinMPM:
    def __init__(self, scene):
        sc = scene
        f8 = lambda a: np.array(a, dtype=np.float64)
        self.n_e, self.n_t, self.n_v = sc.n_elements, sc.n_traditional, sc.n_vertices
        self.n_p = sc.n_particles
        self.n_nv = self.n_p - self.n_v
        self.G, self.grid_lim = sc.n_grid, float(sc.grid_lim)
        self.dx = np.float64(np.float32(sc.grid_lim / sc.n_grid))
        self.inv_dx = np.float64(np.float32(float(sc.n_grid / sc.grid_lim)))
        self.x, self.v = f8(sc.x), f8(sc.v)
        self.C = np.zeros((self.n_p, 3, 3))
        self.F = np.tile(np.eye(3), (self.n_nv, 1, 1))
        self.F_trial = self.F.copy()
        self.stress = np.zeros((self.n_nv, 3, 3))
        self.d, self.R_inv = f8(sc.d), f8(sc.R_inv)
        self.faces = np.asarray(sc.faces, np.int64)
        self.vol = f8(sc.vol)
        p = sc.params
        self.material = MATERIALS[p.get("material", "jelly")]
        self.g = f8(p.get("g", [0, 0, 0]))
        self.rpic_damping = float(np.float32(p.get("rpic_damping", 0.0)))
        self.grid_v_damping_scale = float(np.float32(p.get("grid_v_damping_scale", 1.1)))
        self.density = np.full(self.n_p, float(sc.density))
        self.mass = f8((self.density.astype(np.float32) * sc.vol.astype(np.float32)))
        E, nu = np.float32(sc.E), np.float32(sc.nu)
        self.mu = np.full(self.n_p, np.float64(E / (np.float32(2) * (np.float32(1) + nu))))
        self.lam = np.full(self.n_p, np.float64(E * nu / ((np.float32(1) + nu) * (np.float32(1) - np.float32(2) * nu))))
        self.gamma, self.kappa = float(sc.gamma), float(sc.kappa)
        ang = p.get("friction_angle", 0.0)
        sin_phi = math.sin(ang / 180.0 * 3.14159265)
        self.friction_coeff = float(np.float32(math.tan(ang / 180.0 * 3.14159265)))
        self.alpha = float(np.float32(math.sqrt(2.0 / 3.0) * 2.0 * sin_phi / (3.0 - sin_phi)))
        self.mesh_faces = None if sc.mesh_faces is None else np.asarray(sc.mesh_faces, np.int64)
        self.mesh_x = None if sc.mesh_vertices is None else f8(sc.mesh_vertices)
        self.mesh_v = None if sc.mesh_vertices is None else np.zeros_like(self.mesh_x)
        self.mesh_friction = float(np.float32(sc.mesh_friction))
        self.has_collider = sc.mesh_vertices is not None
        self.has_mover = sc.num_joint_v > 0 or sc.num_joint_f > 0
        self.njv, self.njf = sc.num_joint_v, sc.num_joint_f
        self.bcs = [(k, dict(v)) for k, v in sc.bcs]   # own copies: a moving cuboid updates its point
        self.time = 0.0
        G3 = self.G ** 3
        self.grid_m = np.zeros(G3)
        self.grid_v_in = np.zeros((G3, 3))
        self.grid_v_out = np.zeros((G3, 3))

    # ------------------------------------------------------------------ helpers
    def _nodes(self, base):
        G = self.G
        n = base[:, None, :] + _IJK[None]
        return (n[..., 0] * G + n[..., 1]) * G + n[..., 2]  # [n,27]

    @staticmethod
    def _w27(w):
        return w[:, 0, _IJK[:, 0]] * w[:, 1, _IJK[:, 1]] * w[:, 2, _IJK[:, 2]]

    def _dw27(self, w, dw):
        i, j, k = _IJK[:, 0], _IJK[:, 1], _IJK[:, 2]
        return np.stack([dw[:, 0, i] * w[:, 1, j] * w[:, 2, k], w[:, 0, i] * dw[:, 1, j] * w[:, 2, k],
                         w[:, 0, i] * w[:, 1, j] * dw[:, 2, k]], -1) * self.inv_dx  # [n,27,3]

    # ------------------------------------------------------------------ constitutive
    def cloth_return_mapping(self):
        d = self.d
        Q, R = qr_gs(d)
        R = R.copy()
        r22 = R[:, 2, 2].copy()
        over = r22 > 1.0
        fn = self.kappa * (1.0 - r22) ** 2
        ff = self.gamma * np.sqrt(R[:, 0, 2] ** 2 + R[:, 1, 2] ** 2)
        slide = (~over) & (ff > self.friction_coeff * fn)
        scale = np.where(slide, self.friction_coeff * fn / np.where(ff > 0, ff, 1.0), 1.0)
        R[:, 0, 2] *= scale
        R[:, 1, 2] *= scale
        R[over, 2, 2] = 1.0
        d3 = np.einsum("nij,nj->ni", Q, R[:, :, 2])
        nd = d.copy()
        nd[:, :, 2] = d3
        self.d = nd

    def cloth_stress(self):
        d, n_e = self.d, self.n_e
        iD11, iD12, iD22 = self.R_inv[:, 0], self.R_inv[:, 1], self.R_inv[:, 2]
        mu, lam = self.mu[:n_e], self.lam[:n_e]
        Q, R = qr_gs(d)
        F11 = R[:, 0, 0] * iD11
        F12 = R[:, 0, 0] * iD12 + R[:, 0, 1] * iD22
        F22 = R[:, 1, 1] * iD22
        th = np.arctan2(-F12, F11 + F22)  # polar rotation of [[F11,F12],[0,F22]]
        c, s = np.cos(th), np.sin(th)
        Rot = np.stack([np.stack([c, -s], -1), np.stack([s, c], -1)], -2)
        F2 = np.zeros((n_e, 2, 2)); F2[:, 0, 0] = F11; F2[:, 0, 1] = F12; F2[:, 1, 1] = F22
        iFTJ = np.zeros((n_e, 2, 2)); iFTJ[:, 0, 0] = F22; iFTJ[:, 1, 0] = -F12; iFTJ[:, 1, 1] = F11
        J = F11 * F22
        K2 = 2.0 * mu[:, None, None] * (F2 - Rot) + (lam * (J - 1.0))[:, None, None] * iFTJ
        dr = np.zeros((n_e, 3, 3))
        dr[:, 0, 0], dr[:, 0, 1], dr[:, 1, 1] = K2[:, 0, 0], K2[:, 0, 1], K2[:, 1, 1]
        dr[:, 0, 2], dr[:, 1, 2] = self.gamma * R[:, 0, 2], self.gamma * R[:, 1, 2]
        dr[:, 2, 2] = np.where(R[:, 2, 2] > 1.0, 0.0, -self.kappa * (1.0 - R[:, 2, 2]) ** 2)
        RiDT = np.zeros((n_e, 3, 3))
        RiDT[:, 0, 0] = F11; RiDT[:, 1, 0] = F12; RiDT[:, 1, 1] = F22
        RiDT[:, 2, 0], RiDT[:, 2, 1], RiDT[:, 2, 2] = R[:, 0, 2], R[:, 1, 2], R[:, 2, 2]
        K3 = dr @ RiDT
        K3s = np.triu(K3) + np.transpose(np.triu(K3, 1), (0, 2, 1))
        P = Q @ K3s @ np.linalg.inv(RiDT)
        vol = self.vol[:n_e]
        f2 = -vol[:, None] * (iD11[:, None] * P[:, :, 0] + iD12[:, None] * P[:, :, 1])
        f3 = -vol[:, None] * iD22[:, None] * P[:, :, 1]
        f1 = -(f2 + f3)
        vf = np.zeros((self.n_v, 3))
        np.add.at(vf, self.faces[:, 0], f1)
        np.add.at(vf, self.faces[:, 1], f2)
        np.add.at(vf, self.faces[:, 2], f3)
        self.vertex_force = vf
        self.stress[:n_e] = vol[:, None, None] * np.einsum("ni,nj->nij", P[:, :, 2], d[:, :, 2])

    def trad_stress(self):
        sl = slice(self.n_e, self.n_nv)
        Ft = self.F_trial[sl]
        mu, lam = self.mu[sl], self.lam[sl]
        m = self.material
        if m == 2:  # sand_return_mapping
            U, s, Vt = np.linalg.svd(Ft)
            eps = np.log(np.maximum(np.abs(s), 1e-14))
            tr = eps.sum(-1)
            eh = eps - tr[:, None] / 3.0
            ehn = np.linalg.norm(eh, axis=-1)
            dg = ehn + (3.0 * lam + 2.0 * mu) / (2.0 * mu) * tr * self.alpha
            F = Ft.copy()
            a = (dg > 0) & (tr > 0)
            F[a] = (U @ Vt)[a]
            b = (dg > 0) & (tr <= 0)
            H = eps - eh * (dg / np.where(ehn > 0, ehn, 1.0))[:, None]
            Fb = np.einsum("nij,nj,njk->nik", U, np.exp(H), Vt)
            F[b] = Fb[b]
        elif m in (1, 3, 5):
            raise NotImplementedError("twin covers jelly / sand / cloth; plastic metals are pinned by KATs")
        else:
            F = Ft.copy()
        self.F[sl] = F
        J = np.linalg.det(F)
        U, s, Vt = np.linalg.svd(F)
        # proper rotations; for det F > 0 this is the polar rotation
        Rm = U @ Vt
        S = np.zeros_like(F)
        FT = np.transpose(F, (0, 2, 1))
        if m in (0, 5):
            S = 2.0 * mu[:, None, None] * ((F - Rm) @ FT) + (lam * J * (J - 1.0))[:, None, None] * np.eye(3)
        elif m == 2:
            ls = np.log(s).sum(-1)
            c = (2.0 * mu[:, None] * np.log(s) + lam[:, None] * ls[:, None]) / s
            S = np.einsum("nij,nj,njk->nik", U, c, Vt) @ FT
        S = 0.5 * (S + np.transpose(S, (0, 2, 1)))
        self.stress[sl] = S

    # ------------------------------------------------------------------ transfers
    def p2g(self, dt):
        G3 = self.G ** 3
        base, fx, w, dw = stencil(self.x, self.inv_dx)
        W = self._w27(w)
        dW = self._dw27(w, dw)
        nodes = self._nodes(base)
        dpos = (_IJK[None].astype(np.float64) - fx[:, None, :]) * self.dx
        # mpm_utils.py:528-540: C <- (1 - rpic) C + rpic/2 (C - C^T); C = 0 if rpic < -0.001
        r = self.rpic_damping
        Ca = (1.0 - r) * self.C + 0.5 * r * (self.C - np.transpose(self.C, (0, 2, 1)))
        if r < -0.001:
            Ca = np.zeros_like(Ca)
        mv = self.v[:, None, :] + np.einsum("nij,nkj->nki", Ca, dpos)
        mom = (W * self.mass[:, None])[..., None] * mv
        force = np.zeros((self.n_p, 27, 3))
        S = self.stress.copy()
        S[self.n_e:self.n_nv] *= self.vol[self.n_e:self.n_nv, None, None]
        force[:self.n_nv] = -np.einsum("nij,nkj->nki", S, dW[:self.n_nv])
        if self.n_v:
            force[self.n_nv:] = W[self.n_nv:, :, None] * self.vertex_force[:, None, :]
        add = mom + dt * force
        self.grid_m = np.zeros(G3); self.grid_v_in = np.zeros((G3, 3)); self.grid_v_out = np.zeros((G3, 3))
        np.add.at(self.grid_m, nodes.reshape(-1), (W * self.mass[:, None]).reshape(-1))
        np.add.at(self.grid_v_in, nodes.reshape(-1), add.reshape(-1, 3))

    def grid_update(self, dt):
        act = self.grid_m > 1e-15
        self.grid_v_out[act] = self.grid_v_in[act] / self.grid_m[act, None] + dt * self.g
        if self.grid_v_damping_scale < 1.0:  # add_damping_via_grid, mpm_solver.py:373, mpm_utils.py:1162-1174
            self.grid_v_out -= (1.0 - self.grid_v_damping_scale) * self.grid_v_out

    def _splat(self, pts, vals_list):
        G = self.G
        base, fx, w, dw = stencil(pts, self.inv_dx)
        ok = np.all((base >= 0) & (base < G - 3), axis=1)
        base, w = base[ok], w[ok]
        W = self._w27(w)
        nodes = self._nodes(base).reshape(-1)
        outs = []
        weight = np.zeros(G ** 3)
        np.add.at(weight, nodes, W.reshape(-1))
        for vals in vals_list:
            acc = np.zeros((G ** 3, 3))
            np.add.at(acc, nodes, (W[..., None] * vals[ok][:, None, :]).reshape(-1, 3))
            outs.append(acc)
        return weight, outs

    def mesh_collide(self):
        f = self.mesh_faces
        p0, p1, p2 = self.mesh_x[f[:, 0]], self.mesh_x[f[:, 1]], self.mesh_x[f[:, 2]]
        fp = (p0 + p1 + p2) / 3.0
        fv = (self.mesh_v[f[:, 0]] + self.mesh_v[f[:, 1]] + self.mesh_v[f[:, 2]]) / 3.0
        fn = _normalize(np.cross(p1 - p0, p2 - p0))
        weight, (v_in, nrm) = self._splat(fp, [fv, fn])
        act = weight > 1e-15
        v_mesh = v_in[act] / weight[act, None]
        v = self.grid_v_out[act]
        v_rel = v - v_mesh
        n = _normalize(nrm[act])
        nc = (v_rel * n).sum(-1)
        v_proj = v_rel - np.minimum(nc, 0.0)[:, None] * n
        lp = np.linalg.norm(v_proj, axis=-1)
        fr = (nc < 0) & (lp > 1e-20)
        sc = np.maximum(0.0, lp + nc * self.mesh_friction)
        v_fric = np.where(fr[:, None], sc[:, None] * _normalize(v_proj), v_proj)
        self.grid_v_out[act] = v_fric + v_mesh
        self.collider_weight = weight

    def particle_move(self, joint_t_v, joint_v_v, joint_f_v):
        pts, vals = [], []
        if joint_t_v is not None:
            n = joint_t_v.shape[0]
            pts.append(self.x[self.n_nv - n:self.n_nv]); vals.append(joint_t_v)
        pts.append(self.x[self.n_nv:self.n_nv + self.njv]); vals.append(joint_v_v)
        pts.append(self.x[:self.njf]); vals.append(joint_f_v)
        pts, vals = np.concatenate(pts, 0), np.concatenate(vals, 0).astype(np.float64)
        weight, (vel,) = self._splat(pts, [vals])
        act = weight > 1e-15
        self.grid_v_out[act] = vel[act] / weight[act, None]

    def apply_bcs(self, dt):
        G = self.G
        t = np.float32(self.time)
        for kind, kw in self.bcs:
            st, en = kw.get("start_time", 0.0), kw.get("end_time", 999.0)
            inside = bool(t >= np.float32(st) and t < np.float32(en))
            V = self.grid_v_out.reshape(G, G, G, 3)
            if kind == "velocity_cuboid":   # set_velocity_on_cuboid + host-side modify, mpm_solver.py:929-984
                if inside:
                    ii = np.stack(np.meshgrid(np.arange(G), np.arange(G), np.arange(G), indexing="ij"), -1)
                    pos = (ii.astype(np.float32) * np.float32(self.dx)).astype(np.float32)   # node position as the fp32 kernels see it
                    off = np.abs((pos - np.asarray(kw["point"], np.float32)).astype(np.float32))
                    box = (off < np.asarray(kw["size"], np.float32)).all(-1)
                    V[box] = np.asarray(kw["velocity"], np.float64)
                    kw["point"] = [float(np.float32(p) + np.float32(dt) * np.float32(u)) for p, u in zip(kw["point"], kw["velocity"])]
                elif int(kw.get("reset", 0)) == 1 and t < np.float32(en) + np.float32(15.0) * np.float32(dt):
                    V[...] = 0.0
                continue
            if not inside:
                continue
            if kind == "bounding_box":
                pad = 3
                for a in range(3):
                    lo = [slice(None)] * 3; lo[a] = slice(0, pad)
                    hi = [slice(None)] * 3; hi[a] = slice(G - pad, G)
                    sub = V[tuple(lo)][..., a]; sub[sub < 0] = 0.0
                    sub = V[tuple(hi)][..., a]; sub[sub > 0] = 0.0
            elif kind == "surface_collider":
                nrm = np.asarray(kw["normal"], np.float64); nrm = nrm / np.linalg.norm(nrm)
                ii = np.stack(np.meshgrid(np.arange(G), np.arange(G), np.arange(G), indexing="ij"), -1)
                off = ii * self.dx - np.asarray(kw["point"], np.float64)
                below = (off * nrm).sum(-1) < 0
                if kw.get("surface", "sticky") == "cut":   # surface_type 11, mpm_solver.py:614-622
                    z = ii[..., 2] * self.dx
                    keep = below & (z >= 0.4) & (z <= 0.53)
                    V[keep] = V[keep] * np.array([0.3, 0.0, 0.3])
                    V[below & ~keep] = 0.0
                else:
                    V[below] = 0.0  # quirk Q1: every other surface type ends up writing zero
            else:
                raise NotImplementedError(kind)

    def g2p(self, dt):
        base, fx, w, dw = stencil(self.x, self.inv_dx)
        W, dW = self._w27(w), self._dw27(w, dw)
        gv = self.grid_v_out[self._nodes(base)]  # [n,27,3]
        dpos = _IJK[None].astype(np.float64) - fx[:, None, :]
        new_v = (W[..., None] * gv).sum(1)
        new_C = np.einsum("nki,nkj,nk->nij", gv, dpos, W) * (self.inv_dx * 4.0)
        gradv = np.einsum("nki,nkj->nij", gv, dW)
        ne, nnv = self.n_e, self.n_nv
        a_min, a_max = 2.0 * (1.0 / self.inv_dx), self.grid_lim - 2.0 * (1.0 / self.inv_dx)
        self.v[ne:] = new_v[ne:]
        self.x[ne:] = np.clip(self.x[ne:] + dt * new_v[ne:], a_min, a_max)
        self.C = new_C
        self.F_trial[ne:nnv] = (np.eye(3) + dt * gradv[ne:nnv]) @ self.F[ne:nnv]
        if ne:
            vi = self.faces + nnv
            self.v[:ne] = self.v[vi].sum(1) / 3.0
            self.x[:ne] = self.x[vi].sum(1) / 3.0
            d1 = self.x[vi[:, 1]] - self.x[vi[:, 0]]
            d2 = self.x[vi[:, 2]] - self.x[vi[:, 0]]
            d3 = np.einsum("nij,nj->ni", np.eye(3) + dt * gradv[:ne], self.d[:, :, 2])
            self.d = np.stack([d1, d2, d3], -1)

    def step(self, dt, mesh_x=None, mesh_v=None, joint_traditional_v=None, joint_verts_v=None, joint_faces_v=None):
        dt = float(np.float32(dt))
        if mesh_x is not None:
            self.mesh_x = np.array(mesh_x, np.float64)
        if mesh_v is not None:
            self.mesh_v = np.array(mesh_v, np.float64)
        if self.n_e:
            self.cloth_return_mapping()
            self.cloth_stress()
        else:
            self.vertex_force = np.zeros((self.n_v, 3))
        if self.n_t:
            self.trad_stress()
        self.p2g(dt)
        self.grid_update(dt)
        if self.has_collider:
            self.mesh_collide()
        if self.has_mover and joint_verts_v is not None and joint_faces_v is not None:
            self.particle_move(joint_traditional_v, np.asarray(joint_verts_v), np.asarray(joint_faces_v))
        self.apply_bcs(dt)
        self.g2p(dt)
        self.time += dt
