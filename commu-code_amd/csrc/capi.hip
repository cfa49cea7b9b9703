// library identification + entry points that fan out over the translation units
#include "commu_hip.h"
#include <hip/hip_runtime.h>
extern "C" const char* commu_hip_version(void) { return "commu_hip 0.1 (gfx950)"; }

int commu_seed_salt_elementwise(const unsigned*, hipStream_t);
int commu_seed_salt_relattn(const unsigned*, hipStream_t);
int commu_seed_salt_gemm(const unsigned*, hipStream_t);
int commu_seed_salt_gemm8(const unsigned*, hipStream_t);
int commu_seed_salt_gemm_fp8(const unsigned*, hipStream_t);

extern "C" int commu_set_seed_salt(const unsigned* src, hipStream_t stream) {
    int rc = commu_seed_salt_elementwise(src, stream);
    if (rc == 0) rc = commu_seed_salt_relattn(src, stream);
    if (rc == 0) rc = commu_seed_salt_gemm(src, stream);
    if (rc == 0) rc = commu_seed_salt_gemm8(src, stream);
    if (rc == 0) rc = commu_seed_salt_gemm_fp8(src, stream);
    return rc;
}
