// pk_cell.h - per-(row, unit) arithmetic of the recurrent cells, shared by the
// step-wise and the persistent kernels so both evaluate the same expressions.
//
// Equations follow the reference time loops:
//   liGRU       neural_networks.py:1133-1136
//   RNN         neural_networks.py:1441-1442
//   LSTM        neural_networks.py:460-464
//   GRU         neural_networks.py:632-636
//   minimalGRU  neural_networks.py:1294-1297
//
// Notation: p[g] = input projection of gate g after the BatchNorm affine (or
// +bias), u[g] = h_{t-1}.U_g^T from the recurrent GEMM, m = drop mask value
// (same for all t), hp/cp = previous hidden / cell state.
//
// Saved-for-backward slots (S, NS per cell):
//   liGRU  : z, a                RNN : a
//   LSTM   : f, i, o, g, c       GRU : z, r, a, rh      minGRU : z, a, zh
#pragma once
#include "pk_common.h"

__host__ __device__ constexpr int pk_cell_gates(int cell) {
    return cell == PK_CELL_LIGRU ? 2 : cell == PK_CELL_RNN ? 1 : cell == PK_CELL_LSTM ? 4 : cell == PK_CELL_GRU ? 3 : 2;
}
__host__ __device__ constexpr int pk_cell_saved(int cell) {
    return cell == PK_CELL_LIGRU ? 2 : cell == PK_CELL_RNN ? 1 : cell == PK_CELL_LSTM ? 5 : cell == PK_CELL_GRU ? 4 : 3;
}
// cells whose candidate GEMM depends on a gate of the same step (two GEMM phases per step)
__host__ __device__ constexpr bool pk_cell_two_phase(int cell) { return cell == PK_CELL_GRU || cell == PK_CELL_MINGRU; }

#ifdef PK_CELL_FAST_MATH
__device__ __forceinline__ float pk_sig(float x) { return pk_fast_sigmoid(x); }
#else
__device__ __forceinline__ float pk_sig(float x) { return 1.0f / (1.0f + expf(-x)); }
#endif

// ---------------------------------------------------------------- forward ----
// single-phase cells.  pre[g] = p[g] + u[g].  Writes h (and c), fills s[0..NS).
template <int CELL>
__device__ __forceinline__ void pk_cell_fwd(int act, const float* pre, float hp, float cp, float m, float& h, float& c,
                                            float* s) {
    if (CELL == PK_CELL_LIGRU) {
        const float z = pk_sig(pre[0]);
        const float a = pre[1];
        const float cand = pk_act(act, a) * m;
        h = z * hp + (1.f - z) * cand;
        c = 0.f;
        s[0] = z;
        s[1] = a;
    } else if (CELL == PK_CELL_RNN) {
        const float a = pre[0];
        h = pk_act(act, a) * m;
        c = 0.f;
        s[0] = a;
    } else if (CELL == PK_CELL_LSTM) {
        const float f = pk_sig(pre[0]), i = pk_sig(pre[1]), o = pk_sig(pre[2]);
        const float g = pk_act(act, pre[3]);
        c = i * g * m + f * cp;
        h = o * pk_act(act, c);
        s[0] = f;
        s[1] = i;
        s[2] = o;
        s[3] = g;
        s[4] = c;
    }
}

// two-phase cells, phase 1: gates that depend on h_{t-1} only; returns the vector fed to U_h
template <int CELL>
__device__ __forceinline__ float pk_cell_fwd_p1(const float* pre, float hp, float* s) {
    if (CELL == PK_CELL_GRU) {
        const float z = pk_sig(pre[0]), r = pk_sig(pre[1]);
        s[0] = z;
        s[1] = r;
        const float rh = r * hp;
        s[3] = rh;
        return rh;
    } else {  // minimalGRU
        const float z = pk_sig(pre[0]);
        s[0] = z;
        const float zh = z * hp;
        s[2] = zh;
        return zh;
    }
}
// phase 2: a = p_a + (gate*h).U_h^T
template <int CELL>
__device__ __forceinline__ float pk_cell_fwd_p2(int act, float a, float z, float hp, float m) {
    return z * hp + (1.f - z) * (pk_act(act, a) * m);
}

// --------------------------------------------------------------- backward ----
// single-phase cells.  dh = dL/dh_t (upper layer + carry), dc = carried dL/dc_t.
// Outputs: dg[0..G) = gradients w.r.t. the gate pre-activations (these are the
// dP entries and the operand of the carry GEMM), dh_direct = part of
// dL/dh_{t-1} that does not go through U, dc_prev.
template <int CELL>
__device__ __forceinline__ void pk_cell_bwd(int act, const float* s, float hp, float cp, float m, float dh, float dc,
                                            float* dg, float& dh_direct, float& dc_prev) {
    if (CELL == PK_CELL_LIGRU) {
        const float z = s[0], a = s[1];
        const float cand = pk_act(act, a) * m;
        dg[0] = dh * (hp - cand) * z * (1.f - z);
        dg[1] = dh * (1.f - z) * m * pk_act_grad_from_in(act, a);
        dh_direct = dh * z;
        dc_prev = 0.f;
    } else if (CELL == PK_CELL_RNN) {
        dg[0] = dh * m * pk_act_grad_from_in(act, s[0]);
        dh_direct = 0.f;
        dc_prev = 0.f;
    } else if (CELL == PK_CELL_LSTM) {
        const float f = s[0], i = s[1], o = s[2], g = s[3], c = s[4];
        const float tc = pk_act(act, c);
        const float dct = dc + dh * o * pk_act_grad_from_out(act, tc);
        dg[0] = dct * cp * f * (1.f - f);
        dg[1] = dct * g * m * i * (1.f - i);
        dg[2] = dh * tc * o * (1.f - o);
        dg[3] = dct * i * m * pk_act_grad_from_out(act, g);
        dh_direct = 0.f;
        dc_prev = dct * f;
    }
}

// two-phase cells, backward phase A: da (operand of q = da.U_h) and the z-gate part
// that does not need q.  Returns da; dz_part = dh*(hp - cand)*z(1-z); dh_direct = dh*z.
template <int CELL>
__device__ __forceinline__ float pk_cell_bwd_pa(int act, const float* s, float hp, float m, float dh, float& dz_part,
                                                float& dh_direct) {
    const float z = s[0];
    const float a = (CELL == PK_CELL_GRU) ? s[2] : s[1];
    const float cand = pk_act(act, a) * m;
    dz_part = dh * (hp - cand) * z * (1.f - z);
    dh_direct = dh * z;
    return dh * (1.f - z) * m * pk_act_grad_from_in(act, a);
}
// backward phase B with q = (da.U_h)[unit]: finishes the gate gradients and the
// direct carry.  GRU: dg = [dz, dr, da]; minGRU: dg = [dz, da].
template <int CELL>
__device__ __forceinline__ void pk_cell_bwd_pb(const float* s, float hp, float q, float da, float dz_part, float* dg,
                                               float& dh_direct) {
    if (CELL == PK_CELL_GRU) {
        const float r = s[1];
        dg[0] = dz_part;
        dg[1] = q * hp * r * (1.f - r);
        dg[2] = da;
        dh_direct += q * r;
    } else {
        const float z = s[0];
        dg[0] = dz_part + q * hp * z * (1.f - z);
        dg[1] = da;
        dh_direct += q * z;
    }
}

// per-step LayerNorm of h_t in the persistent recurrences: what the entry points hand to the launchers (null pointer to
// the struct = a layer without it)
struct PkLnHost {
    const float *gamma, *beta;  // [H]
    float eps;
    float* LNS;     // saved for backward: >= pk_rec_ln_saved_floats floats
    float* lnwork;  // scratch: >= pk_rec_ln_work_floats floats
    float *dgamma, *dbeta;  // backward only
};
