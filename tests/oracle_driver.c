/* tests/oracle_driver.c -- TEST INFRASTRUCTURE: runs the CPU oracle (oracle/hotpath_oracle.c) on a stream
 * by parsing it with the product's host parser and passing the resulting plan view across -- the same
 * seam the HIP kernels sit behind. */
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include "../include/j40hip.h"

uint32_t oracle_decode_vardct(const j40hip_vardct_view *v, uint8_t *rgba, float *coeffs_out);
uint32_t oracle_decode_modular(const j40hip_modular_view *v, uint8_t *rgba);
uint32_t oracle_decode_vardct_xyb(const j40hip_vardct_view *v, uint8_t *rgba, float *coeffs_out, float *xyb_out);
void oracle_xyb_to_rgba(const j40hip_vardct_view *v, const float *xyb, uint8_t *rgba);

/* the restoration filters' two ends through the oracle: the XYB samples of a VarDCT stream as the inverse transforms leave them (three
 * planes of width * height floats) and, separately, the colour conversion of such planes with the stream's colour parameters */
__attribute__((visibility("default"))) uint32_t oracle_run_xyb(const void *buf, size_t size, float *xyb_out) {
	uint32_t err = 0;
	j40hip_vardct_view v;
	j40hip_frame *f = j40hip_frame_parse(buf, size, 1, &err);
	if (!f) return err;
	err = j40hip_frame_vardct_view(f, &v);
	if (!err) err = oracle_decode_vardct_xyb(&v, NULL, NULL, xyb_out);
	j40hip_frame_free(f);
	return err;
}
__attribute__((visibility("default"))) uint32_t oracle_colour(const void *buf, size_t size, const float *xyb, uint8_t *rgba) {
	uint32_t err = 0;
	j40hip_vardct_view v;
	j40hip_frame *f = j40hip_frame_parse(buf, size, 1, &err);
	if (!f) return err;
	err = j40hip_frame_vardct_view(f, &v);
	if (!err) oracle_xyb_to_rgba(&v, xyb, rgba);
	j40hip_frame_free(f);
	return err;
}

__attribute__((visibility("default"))) uint32_t oracle_run(const void *buf, size_t size, uint8_t *rgba, float *coeffs_out) {
	uint32_t err = 0;
	int64_t info[32];
	j40hip_frame *f = j40hip_frame_parse(buf, size, 1, &err);
	if (!f) return err;
	j40hip_frame_info(f, info);
	if (info[2]) {
		j40hip_modular_view v;
		err = j40hip_frame_modular_view(f, &v);
		if (!err) err = oracle_decode_modular(&v, rgba);
	} else {
		j40hip_vardct_view v;
		err = j40hip_frame_vardct_view(f, &v);
		if (!err) err = oracle_decode_vardct(&v, rgba, coeffs_out);
	}
	j40hip_frame_free(f);
	return err;
}

/* the seam in the other direction: parse -> view -> j40hip_frame_from_vardct_view (a second handle that never saw the
 * bitstream's headers) -> its view -> the oracle. mode 1: decode that second handle on the GPU instead (rgba = host buffer
 * of width * 4 bytes per row). Modes 2 / 3: the same, with the view flattened into the LF-bundle blob and rebuilt from it. */
__attribute__((visibility("default"))) uint32_t seam_roundtrip(const void *buf, size_t size, uint8_t *rgba, int mode) {
	uint32_t err = 0;
	int64_t info[32];
	j40hip_vardct_view v, v2;
	j40hip_frame *f = j40hip_frame_parse(buf, size, 1, &err), *g;
	if (!f) return err;
	err = j40hip_frame_vardct_view(f, &v);
	if (err) { j40hip_frame_free(f); return err; }
	if (mode >= 2) {   /* through the relocatable blob a sharded decode broadcasts (j40hip_frame_lf_bundle): mode 2 oracle, mode 3 GPU */
		size_t need = j40hip_frame_lf_bundle(f, NULL, 0, &err);
		void *blob = need ? malloc(need) : NULL;
		g = NULL;
		if (blob && j40hip_frame_lf_bundle(f, blob, need, &err) == need) g = j40hip_frame_from_lf_bundle(blob, need, &err);
		free(blob);
		mode -= 2;
	} else g = j40hip_frame_from_vardct_view(&v, &err);
	if (g) {
		j40hip_frame_info(g, info);
		if (mode == 1) {
			err = j40hip_frame_upload(g, 0);
			if (!err) err = j40hip_frame_decode_to_host(g, rgba, (size_t) info[0] * 4);
		} else {
			err = j40hip_frame_vardct_view(g, &v2);
			if (!err) err = oracle_decode_vardct(&v2, rgba, NULL);
		}
		j40hip_frame_free(g);
	}
	j40hip_frame_free(f);
	return err;
}
