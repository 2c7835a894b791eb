// bc.hpp -- grid boundary conditions evaluated per node (shared by both back ends).
// Reference: the `collide` closures registered by add_surface_collider (mpm_solver.py:600-655),
// set_velocity_on_cuboid (:950-973), add_bounding_box (:993-1050), enforce_grid_velocity_by_mask
// (:1341-1352).  Node position = index * dx.
#pragma once
#include "ctx.hpp"
#include "mpm_math.hpp"

namespace mpm {

// returns true when v was (re)written
__device__ __forceinline__ bool apply_bc(const BC &bc, V3 &v, int gx, int gy, int gz, int G, float dx, float time,
                                         float dt, size_t dense_index) {
  if (bc.type == BC_GRIDMASK) {
    if (bc.mask[dense_index] >= 1) { v = v3(0, 0, 0); return true; }
    return false;
  }
  bool in_window = time >= bc.start_time && time < bc.end_time;
  if (bc.type == BC_SURFACE) {
    if (!in_window) return false;
    V3 off = v3((float)gx * dx - bc.point[0], (float)gy * dx - bc.point[1], (float)gz * dx - bc.point[2]);
    float dp = off.x * bc.normal[0] + off.y * bc.normal[1] + off.z * bc.normal[2];
    if (!(dp < 0.0f)) return false;
    if (bc.surface_type == 11) {
      float z = (float)gz * dx;
      if (z < 0.4f || z > 0.53f) v = v3(0, 0, 0);
      else v = v3(v.x * 0.3f, 0.0f, v.z * 0.3f);
    } else {
      // sticky; slip / frictional surfaces also end in a zero write (quirk Q1, mpm_solver.py:653-655)
      v = v3(0, 0, 0);
    }
    return true;
  }
  if (bc.type == BC_CUBOID) {
    if (in_window) {
      V3 off = v3((float)gx * dx - bc.point[0], (float)gy * dx - bc.point[1], (float)gz * dx - bc.point[2]);
      if (fabsf(off.x) < bc.size[0] && fabsf(off.y) < bc.size[1] && fabsf(off.z) < bc.size[2]) {
        v = v3(bc.velocity[0], bc.velocity[1], bc.velocity[2]);
        return true;
      }
      return false;
    }
    if (bc.reset == 1 && time < bc.end_time + 15.0f * dt) { v = v3(0, 0, 0); return true; }
    return false;
  }
  if (bc.type == BC_BBOX) {
    if (!in_window) return false;
    const int padding = 3;
    bool ch = false;
    if (gx < padding && v.x < 0.0f) { v.x = 0.0f; ch = true; }
    if (gx >= G - padding && v.x > 0.0f) { v.x = 0.0f; ch = true; }
    if (gy < padding && v.y < 0.0f) { v.y = 0.0f; ch = true; }
    if (gy >= G - padding && v.y > 0.0f) { v.y = 0.0f; ch = true; }
    if (gz < padding && v.z < 0.0f) { v.z = 0.0f; ch = true; }
    if (gz >= G - padding && v.z > 0.0f) { v.z = 0.0f; ch = true; }
    return ch;
  }
  return false;
}

// Conservative box test for apply_bc: false only if bc cannot change any node with indices in [lo, hi] (per axis) at this
// time.  g2p evaluates nodes tile by tile and skips the BC loop for tiles no BC can reach.
__device__ __forceinline__ bool bc_may_touch(const BC &bc, int lox, int loy, int loz, int hix, int hiy, int hiz, int G,
                                             float dx, float time, float dt) {
  if (bc.type == BC_GRIDMASK) return true;
  bool in_window = time >= bc.start_time && time < bc.end_time;
  if (bc.type == BC_SURFACE) {
    if (!in_window) return false;
    // dp is affine in the node index: its minimum over the box is taken at a corner (same float ops as apply_bc)
    float lo[3] = {(float)lox * dx - bc.point[0], (float)loy * dx - bc.point[1], (float)loz * dx - bc.point[2]};
    float hi[3] = {(float)hix * dx - bc.point[0], (float)hiy * dx - bc.point[1], (float)hiz * dx - bc.point[2]};
    float dpmin = 0.0f, mag = 0.0f;
    for (int a = 0; a < 3; ++a) {
      dpmin += fminf(lo[a] * bc.normal[a], hi[a] * bc.normal[a]);
      mag += fmaxf(fabsf(lo[a] * bc.normal[a]), fabsf(hi[a] * bc.normal[a]));
    }
    return dpmin < 1e-5f * mag + 1e-30f;  // margin for the different summation order
  }
  if (bc.type == BC_CUBOID) {
    if (!in_window) return bc.reset == 1 && time < bc.end_time + 15.0f * dt;
    // |g dx - p| < size somewhere in [lo, hi]: g dx - p is monotone in g
    return ((float)lox * dx - bc.point[0] < bc.size[0]) && ((float)hix * dx - bc.point[0] > -bc.size[0]) &&
           ((float)loy * dx - bc.point[1] < bc.size[1]) && ((float)hiy * dx - bc.point[1] > -bc.size[1]) &&
           ((float)loz * dx - bc.point[2] < bc.size[2]) && ((float)hiz * dx - bc.point[2] > -bc.size[2]);
  }
  if (bc.type == BC_BBOX) {
    const int padding = 3;
    return in_window && (lox < padding || loy < padding || loz < padding || hix >= G - padding || hiy >= G - padding ||
                         hiz >= G - padding);
  }
  return false;
}

// host-side `modify` of set_velocity_on_cuboid, mpm_solver.py:975-981
inline void bc_host_modify(BC &bc, float time, float dt) {
  if (bc.type == BC_CUBOID && time >= bc.start_time && time < bc.end_time)
    for (int a = 0; a < 3; ++a) bc.point[a] = bc.point[a] + dt * bc.velocity[a];
}

}  // namespace mpm
