/* gv_test_hooks.h -- entry points the library exports for tests/ only.  NOT part of the ABI: include/gridvision_hip.h
 * does not declare them, INTEGRATION.md does not list them, a maintainer's binding never sees them; they may change or
 * go without an ABI version step. */
#ifndef GV_TEST_HOOKS_H
#define GV_TEST_HOOKS_H
#include "../../include/gridvision_hip.h"
#ifdef __cplusplus
extern "C" {
#endif
/* No RCCL, one device: runs the sharded frame (gv_frame_enqueue_sharded's kernels, bands and slices) for EVERY rank of
 * a `world`-GPU job on this handle -- the resident cloud is the whole cloud, rank r takes points
 * [n*r/world, n*(r+1)/world) -- with the exchanges done by device copies; the result must equal gv_process_frame's
 * (tests/test_gpu_parity.py::test_sharded_frame_every_rank_emulated). */
int gv_test_frame_sharded_emulated(gv_handle h, const gv_frame_desc *desc, int32_t world);
#ifdef __cplusplus
}
#endif
#endif
