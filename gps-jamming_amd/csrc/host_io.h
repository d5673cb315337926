// Host-side plumbing shared by api.hip, host_io.hip and comm.hip: the waits (never under the context lock) and the
// per-call lanes of the host-buffer entry points.
#pragma once
#include "gj_common.h"

namespace gj {

int wait_stream(gj_ctx* ctx, hipStream_t stream);
int wait_event(gj_ctx* ctx, hipEvent_t ev);
hipStream_t current_stream(gj_ctx* ctx);

gj_lane* lane_checkout(gj_ctx* ctx);          // nullptr + last error when a lane cannot be made
void lane_checkin(gj_ctx* ctx, gj_lane* lane);
void lane_free(gj_lane* lane);                // gj_destroy only
int lane_sweep(gj_ctx* ctx);                  // lanes of ended callers taken back (drained, stripped, free again)

void comm_detach_all(gj_ctx* ctx);            // comm.hip

}   // namespace gj
