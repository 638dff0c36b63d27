/* tests/oracle_driver.c -- TEST INFRASTRUCTURE: runs the CPU oracle (oracle/hotpath_oracle.c) on a stream
 * by parsing it with the product's host parser and passing the resulting plan view across -- the same
 * seam the HIP kernels sit behind. */
#include <stdint.h>
#include <stddef.h>
#include "../include/j40hip.h"

uint32_t oracle_decode_vardct(const j40hip_vardct_view *v, uint8_t *rgba, float *coeffs_out);
uint32_t oracle_decode_modular(const j40hip_modular_view *v, uint8_t *rgba);

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
