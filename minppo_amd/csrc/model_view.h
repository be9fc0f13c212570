// ModelView: the robot model as the physics kernel sees it (dims + device pointers into the
// blob), and the per-environment LDS layout derived from the dims.
// Blob layout: minppo_amd/model.py (_to_blob).  Array order must match _BLOB_INT + _BLOB_F32.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace mppo {

enum BlobInt {
  BI_body_parent, BI_body_rootid, BI_body_depth, BI_body_jntadr, BI_body_jntnum, BI_body_dofadr, BI_body_dofnum,
  BI_jnt_type, BI_jnt_qposadr, BI_jnt_dofadr, BI_jnt_bodyid, BI_jnt_limited,
  BI_dof_bodyid, BI_dof_jntid, BI_dof_parentid,
  BI_act_dofid, BI_act_qposadr, BI_act_ctrllimited, BI_act_forcelimited,
  BI_con_bodyid, BI_lim_jntid, BI_pair_body,
  BI_level_adr, BI_level_body, BI_root_body, BI_body_subtree_mask, BI_body_ancdof_mask, BI_dof_velmask, BI_dof_qposadr,
  BI_con_cvx, BI_cvx_body, BI_cvx_vadr,
  BI_COUNT
};
enum BlobF32 {
  BF_gravity = BI_COUNT, BF_body_pos, BF_body_quat, BF_body_ipos, BF_body_iquat, BF_body_mass, BF_body_inertia,
  BF_jnt_pos, BF_jnt_axis, BF_jnt_range, BF_jnt_stiffness,
  BF_dof_armature, BF_dof_damping, BF_dof_invweight0, BF_body_invweight0,
  BF_qpos0, BF_qpos_spring,
  BF_act_gear, BF_act_gain, BF_act_bias, BF_act_ctrlrange, BF_act_forcerange,
  BF_con_lpos, BF_con_radius, BF_con_friction, BF_con_axis, BF_pair_geom,
  BF_contact_solref, BF_contact_solimp, BF_limit_solref, BF_limit_solimp,
  BF_cvx_vert, BF_dof_actfrcrange,
  BLOB_ARRAY_COUNT
};

constexpr uint32_t kBlobMagic = 0x4D50504F;
constexpr uint32_t kBlobVersion = 7;
constexpr int kBlobHeaderWords = 64;
constexpr int JNT_FREE = 0, JNT_HINGE = 2, JNT_SLIDE = 3;
constexpr float MJ_MINVAL = 1e-15f, MJ_MINIMP = 0.0001f, MJ_MAXIMP = 0.9999f;

typedef unsigned long long u64;

// Dims + where the blob lives.  The kernel copies the blob into LDS once per workgroup and reads every table
// from there: o[k] is the word offset of array k (BlobInt / BlobF32 index) inside the blob.
struct ModelView {
  int nq, nv, nu, nbody, njnt, ncon, nlimit, nefc, iterations, ls_iterations, nlevel, nroot;
  int npair;  // the last npair of the ncon contact slots are geom-geom pairs (pair_body / pair_geom); the others are ground contacts
  int ncvx, ncvxvert;  // convex (mesh) geoms against the plane: four contact slots each (con_cvx), ncvxvert hull vertices in all (cvx_vert, body frame)
  int obs_dim, obs_pad, rec_dim;
  int include_c;  // observation = qpos, qvel, cinert[1:], cvel[1:], qfrc_actuator (1) or qpos, qvel, qfrc_actuator (0): env.py:246-259
  float timestep, tolerance, ls_tolerance, impratio, plane_z, meaninertia;
  const int* blob;   // device copy of the whole blob (16-byte aligned)
  int blob_words;    // multiple of 4: the TABLE part (what the kernel copies into LDS)
  int epw;           // environments per wave of the run-time-sized kernel: 4, or 2 / 1 for a robot whose working set would not fit LDS four at a time
  int ncyl;          // cylinders against the plane: three contact slots each (con_cvx = -2, -3, -4); run-time-sized kernel only
  int hull_words;    // the hull section behind it (0: the model has no convex geom in a geom-geom pair); read from global memory
  int o[BLOB_ARRAY_COUNT];
};

// The dims every array length follows from, and the canonical placement of the arrays (model.py _to_blob: directory order, each
// array padded to 4 words, first array right after the directory).  mppo_model_open refuses a blob laid out differently, so a
// kernel compiled for fixed dims may take the offsets as constants.
struct BlobDims { int nq, nv, nu, nbody, njnt, ncon, nlimit, npair, nlevel, nroot, ncvx, ncvxvert;
                  int hull, ncyl; };  // (hull: 1 if the model has a hull section; ncyl: its cylinders - they select code, not table sizes)
struct BlobOffsets { int o[BLOB_ARRAY_COUNT]; int words; };
__host__ __device__ constexpr inline int blob_array_len(const BlobDims& d, int k) {
  switch (k) {
    case BI_body_parent: case BI_body_rootid: case BI_body_depth: case BI_body_jntadr: case BI_body_jntnum: case BI_body_dofadr: case BI_body_dofnum:
    case BF_body_mass: return d.nbody;
    case BI_jnt_type: case BI_jnt_qposadr: case BI_jnt_dofadr: case BI_jnt_bodyid: case BI_jnt_limited: case BF_jnt_stiffness: return d.njnt;
    case BI_dof_bodyid: case BI_dof_jntid: case BI_dof_parentid: case BI_dof_qposadr: case BF_dof_armature: case BF_dof_damping: case BF_dof_invweight0: return d.nv;
    case BI_act_dofid: case BI_act_qposadr: case BI_act_ctrllimited: case BI_act_forcelimited: case BF_act_gear: case BF_act_gain: return d.nu;
    case BI_con_bodyid: case BF_con_radius: case BI_con_cvx: return d.ncon;
    case BI_cvx_body: return d.ncvx;
    case BI_cvx_vadr: return d.ncvx + 1;
    case BF_cvx_vert: return 3 * d.ncvxvert;
    case BI_lim_jntid: return d.nlimit;
    case BI_pair_body: return 2 * d.npair;
    case BI_level_adr: return d.nlevel + 1;
    case BI_level_body: return d.nbody - 1;
    case BI_root_body: return d.nroot;
    case BI_body_subtree_mask: return (d.nbody > 64 ? 4 : 2) * d.nbody;  // a 64-bit word per body: bodies 0..63 of its subtree; beyond 64 bodies a second block of words: bodies 64..127
    case BI_body_ancdof_mask: case BF_body_invweight0: return 2 * d.nbody;
    case BI_dof_velmask: case BF_dof_actfrcrange: return 2 * d.nv;
    case BF_gravity: return 3;
    case BF_body_pos: case BF_body_ipos: case BF_body_inertia: return 3 * d.nbody;
    case BF_body_quat: case BF_body_iquat: return 4 * d.nbody;
    case BF_jnt_pos: case BF_jnt_axis: return 3 * d.njnt;
    case BF_jnt_range: return 2 * d.njnt;
    case BF_qpos0: case BF_qpos_spring: return d.nq;
    case BF_act_bias: return 3 * d.nu;
    case BF_act_ctrlrange: case BF_act_forcerange: return 2 * d.nu;
    case BF_con_lpos: case BF_con_friction: case BF_con_axis: return 3 * d.ncon;
    case BF_pair_geom: return 16 * d.npair;
    case BF_contact_solref: case BF_limit_solref: return 2;
    case BF_contact_solimp: case BF_limit_solimp: return 5;
    default: return -1;
  }
}
__host__ __device__ constexpr inline BlobOffsets blob_offsets(const BlobDims& d) {
  BlobOffsets b{};
  int cur = (64 + 2 * (int)BLOB_ARRAY_COUNT + 3) & ~3;  // kBlobHeaderWords + directory
  for (int k = 0; k < BLOB_ARRAY_COUNT; ++k) {
    b.o[k] = cur;
    cur += (blob_array_len(d, k) + 3) & ~3;
  }
  b.words = cur;
  return b;
}

// The hull section (model.py _hull_section): the convex geoms (boxes, mesh hulls) that meet spheres / capsules of other bodies, as MJX's
// sphere_convex / capsule_convex want them - per hull a range of vertices, of polygon faces (vertex index lists, counter-clockwise seen
// from outside; outward unit normals) and of edges (vertex pair + the normals of the two faces beside it), everything in the frame of the
// body the geom is fixed to.  Eight header words (nhull, nvert, nface, nfidx, nedge, 0, 0, 0), then the arrays in this order, each padded
// to 4 words; offsets below are in words from the section's start.  (Header word 5: the number of edge directions, blob version 7.)
struct HullView {
  int nhull, nvert, nface, nfidx, nedge;
  int vadr, fadr, eadr, face_adr, fidx, edge, vert, fnormal, enormal;
  int words;
  int nudir, udadr, udir;  // round 6 (blob 7): per hull a range of unit edge DIRECTIONS, the parallel ones dropped - the edge axes of convex_convex
};
__host__ __device__ constexpr inline HullView hull_view(int nhull, int nvert, int nface, int nfidx, int nedge, int nudir = 0) {
  HullView h{nhull, nvert, nface, nfidx, nedge, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, nudir, 0, 0};
  int cur = 8;
  auto take = [&](int n) { const int o = cur; cur += (n + 3) & ~3; return o; };
  h.vadr = take(nhull + 1); h.fadr = take(nhull + 1); h.eadr = take(nhull + 1); h.face_adr = take(nface + 1); h.fidx = take(nfidx);
  h.edge = take(2 * nedge); h.vert = take(3 * nvert); h.fnormal = take(3 * nface); h.enormal = take(6 * nedge);
  h.udadr = take(nhull + 1); h.udir = take(3 * nudir);
  h.words = cur;
  return h;
}

// Per-environment LDS layout (offsets in floats).  What is alive from one end of a step to the other sits in front; region "A" is
// time-shared by four lifetimes that follow one another (each separated from the next by a synchronisation point of the kernel):
//   A1 kinematics: poses (xpos, xquat, xipos), joint anchors / axes (the rotation matrices of bodies and inertial frames are recomputed
//      from the quaternions where they are used - the same arithmetic, nothing stored)          - until the contact frames are built
//   A2 factorisation work copy: a [nv][ldc] square (<= 32 dofs, fixed-size kernel: rows in registers; also the square through which the
//      inverse factor is transposed into the row owners' registers) or one packed lower triangle (run-time-sized kernel: M, and at the
//      end of the step M + h D)
//   A3 velocity / RNE scratch: cdofdot, cfrc, cvel                                             - until qfrc_bias is done
//   A4 the contact rows of the constraint Jacobian (joint-limit rows are a signed row number per dof: dlim, never a dense row)
// A second time-shared region, B, holds cdof + contact geometry, then the solver's vectors.  cinert (computed after the kinematics, last
// read by RNE) shares the storage of the constraint rows' vectors D / aref / jaref / jv, which are born in make_constraint, after RNE.
// cinert, cvel and qfrc_actuator leave LDS for the state record as soon as they are final (they are part of the observation, not of the solver).
// `li_regs` (fixed-size kernels up to 32 dofs, round 4): the inverse Cholesky factor lives in registers (a lane's rows and columns); there
// is no LL square, and the factor of M + h D for the Euler step is computed at the END of the step into the same registers instead of
// being kept from the start - 9.5 instead of 12.9 KB per environment for the 26-dof robot = four waves per CU instead of three.
struct PhysLds {
  int qpos, qvel, ctrl, warm;  // ctrl shares t1's storage (dead before the solver; re-read from the action every frame)
  int rootcom, cdof;
  int ldc;         // row stride of the factorisation work copy in a fixed-size kernel: nv rounded up to 4 (rows are read as float4)
  int M, LL, ldm;  // LL (kernels that keep the factor in LDS): the inverse Cholesky factor as a packed lower triangle (row i at i (i + 1) / 2) - of M during the
                   // step, of M + h*D once the second factorisation at the end of the step has run (round 6; round 5 kept both in one square)
  int qfs, qas, qacc, Ma, grad, Mgrad, search, mv, qfc, t0, t1;
  int dlim;        // per dof, an int: +-(row + 1) of its active joint-limit row, the sign being the row's single Jacobian entry (0: none)
  int D, aref, jaref, jv, force;  // force shares jv's storage (jv is dead once the step along the search direction is taken)
  int conpos, condist, confr;  // per contact slot: point, distance, frame rows (normal, first tangent)
  int cvxsel, cvxok;           // per convex geom: the four hull vertices chosen this step (body frame) and whether each slot is a first occurrence
  int A, xpos, xquat, xipos, xanchor, xaxis, C, cdofdot, cfrc, cvel, cinert, J, ldj;
  int total;
  // Round 6 - matrices that leave LDS for a per-environment record in global memory (L2 / Infinity-Cache resident; `spill`, chosen per
  // model by spill_for below so that a CU's 160 KB hold four waves): offsets in floats inside the record, -1 = the matrix is in LDS.
  //   gJc  [ncon][nv][4]       the four pyramid rows of a contact side by side: lane = dof reads one float4 per contact   (J^T f)
  //   gJq  [nvq][4 ncon][4]    four consecutive dofs of a row side by side: lane = row reads one float4 per dof quad      (J x)
  //   gM   [nvq][nv][4]        four consecutive columns of a row side by side: lane = row reads one float4 per quad       (M x)
  // (nvq = ceil(nv / 4); lanes of an environment read consecutive 16-byte words.)  jmask: per contact slot, in LDS, the dofs its rows
  // touch (the ancestors of its one or two bodies) if the contact is active this step, else 0 - writers and readers skip the rest.
  int spill, gJc, gJq, gM, gwords, jmask;
};
constexpr int kSpillJ = 1, kSpillM = 2;

__host__ __device__ constexpr inline int imax_(int a, int b) { return a > b ? a : b; }

__host__ __device__ constexpr inline PhysLds make_phys_lds(int nq, int nv, int nu, int nbody, int njnt, int ncon, int nefc, int nroot, int ncvx = 0, bool li_regs = false, int spill = 0) {
  PhysLds p{};
  int o = 0;
  auto take = [&](int n) { int r = o; o += (n + 3) & ~3; return r; };
  (void)nu;  // (at most nv actuators: mppo_model_open)
  p.qpos = take(nq); p.qvel = take(nv); p.warm = take(nv);
  p.rootcom = take(3 * (nroot > 0 ? nroot : 1));
  p.ldm = nv + 1;  // odd row stride: a column read by 16 lanes hits 16 different banks
  p.spill = ncon > 0 ? spill : (spill & ~kSpillJ);
  p.M = (p.spill & kSpillM) ? 0 : take(nv * p.ldm);
  const int ntri = (nv * (nv + 1) / 2 + 3) & ~3;  // a packed lower triangle: row i at i (i + 1) / 2
  p.LL = li_regs ? p.M : take(ntri);  // (li_regs: never addressed)
  p.qfs = take(nv); p.qas = take(nv); p.t0 = take(nv); p.t1 = take(nv);
  p.ctrl = p.t1;
  p.dlim = take(nv);
  const int ne = nefc > 0 ? nefc : 1;
  const int rows0 = o;
  p.D = take(ne); p.aref = take(ne); p.jaref = take(ne); p.jv = take(ne); p.force = p.jv;
  const int rows1 = o;
  o = rows0; p.cinert = take(10 * nbody); o = imax_(o, rows1);
  p.cvxsel = take(12 * ncvx); p.cvxok = take(4 * ncvx);
  p.jmask = (p.spill & kSpillJ) ? take(2 * ncon) : 0;
  // region B, two lifetimes: the dynamics' cdof and the contact geometry, dead once the Jacobian is built | the solver's nv-vectors,
  // born after that (qacc stays until the end of the step: it is the next step's warm start)
  const int B = o;
  p.cdof = take(6 * nv); p.conpos = take(3 * (ncon > 0 ? ncon : 1)); p.condist = take(ncon > 0 ? ncon : 1); p.confr = take(6 * (ncon > 0 ? ncon : 1));
  const int B1 = o;
  o = B; p.qacc = take(nv); p.Ma = take(nv); p.grad = take(nv); p.Mgrad = take(nv); p.search = take(nv); p.mv = take(nv); p.qfc = take(nv);
  o = imax_(o, B1);
  p.A = o;
  p.xpos = take(3 * nbody); p.xquat = take(4 * nbody); p.xipos = take(3 * nbody); p.xanchor = take(3 * njnt); p.xaxis = take(3 * njnt);
  int end = o;
  p.ldc = (nv + 3) & ~3;
  o = p.A; p.C = take(li_regs ? imax_(nv * imax_(p.ldm, p.ldc), 2 * ntri) : ntri); end = imax_(end, o);
  o = p.A; p.cdofdot = take(6 * nv); p.cfrc = take(6 * nbody); p.cvel = take(6 * nbody); end = imax_(end, o);
  p.ldj = nv + 1;
  o = p.A; p.J = (p.spill & kSpillJ) ? o : take((ncon > 0 ? 4 * ncon : 1) * p.ldj); end = imax_(end, o);
  {
    const int nvq = (nv + 3) / 4;
    int go = 0;
    auto gtake = [&](int n) { int r = go; go += (n + 63) & ~63; return r; };  // (256-byte aligned pieces)
    p.gJc = (p.spill & kSpillJ) ? gtake(ncon * nv * 4) : -1;
    p.gJq = (p.spill & kSpillJ) ? gtake(nvq * 4 * ncon * 4) : -1;
    p.gM = (p.spill & kSpillM) ? gtake(nvq * nv * 4) : -1;
    p.gwords = go;
  }
  // an environment's arrays start 16 banks after its neighbour's (total = 16 mod 64 words): the four environments of a wave read the
  // same logical address at the same time, 16 consecutive words or 16 rows of odd stride each - disjoint sets of the 64 banks
  p.total = ((end + 47) & ~63) + 16;
  return p;
}

// Which matrices leave LDS (PhysLds::spill): the smallest set - none, the contact Jacobian, the Jacobian and M - with which a CU holds
// the most waves, counted up to four (one wave per SIMD: 4096 environments of four per wave are then ONE round of the 256 CUs; a fifth
// wave per CU does not shorten anything the matrices' longer way would not lengthen).  A pure function of the dims: the model-specialised
// kernels evaluate it at compile time, mppo_model_open at run time.
constexpr int kLdsPerCu = 160 * 1024;
__host__ __device__ constexpr inline int waves_per_cu(long long blob_words, long long env_words, int epw) {
  int best = 0;
  for (int w = 1; w <= 4; ++w) {
    const long long bytes = (blob_words + env_words * epw * w) * 4;
    const int per_cu = bytes <= kLdsPerCu ? (int)(kLdsPerCu / bytes) * w : 0;
    if (per_cu > best) best = per_cu;
  }
  return best;
}
__host__ __device__ constexpr inline int spill_for(int nq, int nv, int nu, int nbody, int njnt, int ncon, int nefc, int nroot, int ncvx, bool li_regs, int blob_words) {
  int best = 0, best_w = -1;
  const int opts[3] = {0, kSpillJ, kSpillJ | kSpillM};
  for (int t = 0; t < 3; ++t) {
    const int s = opts[t];
    const int w = waves_per_cu(blob_words, make_phys_lds(nq, nv, nu, nbody, njnt, ncon, nefc, nroot, ncvx, li_regs, s).total, 4);
    const int wc = w > 4 ? 4 : w;
    if (wc > best_w) { best_w = wc; best = s; }
  }
  return best;
}

// Geometry of the environment kernel.  An environment is private to ONE wavefront (16 lanes = one DPP row), so everything
// after the staging of the model tables synchronises at wave level only (program order; no s_barrier).  A wave carries
// kEnvsPerWave environments on its first rows (the other rows retire after the staging); the waves of a workgroup share one LDS
// copy of the model tables.  Measured on MI355X, 4096 envs, synth_stompy_pro (tools/env_iters_probe.py), envs/wave x waves/block:
//   4 x 1: 158 us (default)   2 x 8: 183 us   2 x 4 and 2 x 2: 300 us (LDS lets only one block per CU run)   1 x 4: 600 us
// i.e. the kernel is bound by the instruction stream each WAVE issues (the same ~27 k VALU instructions whether 16 or 64 lanes
// are active), not by latency: two half-filled waves per SIMD run 1.7x faster per wave, but there are twice as many.
#ifndef MPPO_ENVS_PER_WAVE
#define MPPO_ENVS_PER_WAVE 4
#endif
constexpr int kGroupLanes = 16;                      // lanes that cooperate on one environment
#ifndef MPPO_REGCHOL_MAX_NV
#define MPPO_REGCHOL_MAX_NV 48
#endif
constexpr int kRegCholMaxNv = MPPO_REGCHOL_MAX_NV;   // model-specialised kernels up to this many dofs keep the Cholesky factors in registers (three rows per lane)
constexpr int kEnvsPerWave = MPPO_ENVS_PER_WAVE;     // 1, 2 or 4
constexpr int kMaxWavesPerBlock = 4;                 // the number of waves per workgroup is chosen per model (mppo_model_open): the most
                                                     // waves per CU that 160 KB of LDS hold, one copy of the model tables per workgroup

}  // namespace mppo
