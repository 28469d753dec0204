// EAM / ADP kernels (placeholder until the analytic-potential kernels land).
#include <hip/hip_runtime.h>

#include <string>

#include "ta_device.h"

namespace ta {
struct EamModel {};
EamModel *eam_create(const ta_model_desc *, std::string &err) {
  err = "EAM/ADP models are not implemented yet";
  return nullptr;
}
void eam_destroy(EamModel *m) { delete m; }
void eam_ensure(EamModel *, const DeviceBatch &) {}
void eam_compute(EamModel *, const DeviceBatch &, uint32_t, hipStream_t, hipEvent_t *) {}
}  // namespace ta
