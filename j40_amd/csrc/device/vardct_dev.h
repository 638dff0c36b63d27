// j40_amd/csrc/device/vardct_dev.h -- per-varblock geometry and the fused coefficient load
// (dequantise + chroma-from-luma + LLF substitution) shared by the coefficients -> pixels kernels
// (j40__dequant_hf, j40.h:7053; j40__combine_vardct_from_lf_group, j40.h:7099-7175).
#pragma once
#include "idct_dev.h"

namespace j40hip {

struct VbGeom {
	int32_t coeff_base;   // index of the block's first coefficient in plan.coeffs[c]
	int32_t llf_base;     // index of the block's first LLF coefficient in plan.llf[c]
	float mult[3];
	float kx_hf, kb_hf;
	int32_t px, py;       // top-left pixel in the frame
	int32_t effw, effh;   // visible size
};

J40_DEV VbGeom varblock_geometry(const DevPlan &plan, const DevVarblock &vb) {
	const DevFrame &f = *plan.frame;
	VbGeom g;
	g.coeff_base = vb.coeff_base; g.llf_base = vb.llf_base;
	g.mult[1] = vb.mult1; g.mult[0] = vb.mult1 * f.x_qm_mul; g.mult[2] = vb.mult1 * f.b_qm_mul;   // j40.h:7078-7080
	g.kx_hf = f.base_corr_x + f.inv_colour_factor * (float) plan.xfromy[vb.c64];   // j40.h:7138-7143
	g.kb_hf = f.base_corr_b + f.inv_colour_factor * (float) plan.bfromy[vb.c64];
	g.px = vb.px; g.py = vb.py; g.effw = vb.effw; g.effh = vb.effh;
	return g;
}

// loads coefficient `i` (canonical layout index) of all three channels: dequantised, chroma-from-luma
// applied, LLF corner substituted (j40.h:7086-7094, 7157-7172)
J40_DEV void load_coeff3(const DevPlan &plan, const VbGeom &g, const float *dq, int32_t dq_size, int32_t i, int32_t long_side, int32_t vh8, int32_t vw8, float out[3], const uint16_t *inv_order = nullptr) {
	const DevFrame &f = *plan.frame;
	const int32_t srow = i / long_side, scol = i - srow * long_side;
	if (srow < vh8 && scol < vw8) {
		const int32_t l = g.llf_base + srow * vw8 + scol;
		const float lx = plan.llf[0][l], ly = plan.llf[1][l], lb = plan.llf[2][l];
		out[0] = lx + ly * f.kx_lf; out[1] = ly; out[2] = lb + ly * f.kb_lf;
		return;
	}
	// scan-order storage (single-pass frames): canonical index i lives at scan position inv_order[c][i]
	const int32_t ix = inv_order ? inv_order[i] : i, iy = inv_order ? inv_order[dq_size + i] : i, ib = inv_order ? inv_order[2 * dq_size + i] : i;
	const float cx = plan.coeffs[0][g.coeff_base + ix], cy = plan.coeffs[1][g.coeff_base + iy], cb = plan.coeffs[2][g.coeff_base + ib];
	if (plan.clear_after_read) {   // leave the planes all-zero for the next decode
		if (cx != 0.0f) plan.coeffs[0][g.coeff_base + ix] = 0.0f;
		if (cy != 0.0f) plan.coeffs[1][g.coeff_base + iy] = 0.0f;
		if (cb != 0.0f) plan.coeffs[2][g.coeff_base + ib] = 0.0f;
	}
	const float qx = dequant_coeff(cx, f.quant_bias[0], f.quant_bias_num, g.mult[0], dq[i]);
	const float qy = dequant_coeff(cy, f.quant_bias[1], f.quant_bias_num, g.mult[1], dq[dq_size + i]);
	const float qb = dequant_coeff(cb, f.quant_bias[2], f.quant_bias_num, g.mult[2], dq[2 * dq_size + i]);
	out[0] = qx + qy * g.kx_hf; out[1] = qy; out[2] = qb + qy * g.kb_hf;
}


} // namespace j40hip
