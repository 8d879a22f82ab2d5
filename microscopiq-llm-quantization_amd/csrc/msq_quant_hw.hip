// msq_quant_hw.hip -- fake-quant kernels whose element codec is the gfx950 scaled converts
// (v_cvt_scalef32_pk_{fp4,fp8,bf8}_f32 and back); see outlier_block_fast<..., HW> in msq_outlier_core.h.
// Own translation unit: the kernels are heavy templates (5 block sizes x 2 layouts x 2 modes).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/msq.h"
#include "msq_device.h"
#include "msq_host.h"

using namespace msq;

#include "msq_outlier_kernels.h"

extern "C" void msq_set_error_(const char* msg);

// mode 1: inliers and outliers through the converts; mode 2: inliers through the converts, posit outliers
// dtype 0 = f32, 1 = fp16 tensors, 2 = bf16 tensors (both computed in fp32)
extern "C" int msq_launch_outlier_hw_(const void* in, void* out, const OutlierArgs* A, int block, int mode, int dtype, void* stream) {
    bool ok;
    if (dtype == 1) ok = (mode == 1) ? launch_outlier_variant<3, f16io_t>(in, out, *A, block, (hipStream_t)stream)
                                     : launch_outlier_variant<4, f16io_t>(in, out, *A, block, (hipStream_t)stream);
    else if (dtype == 2) ok = (mode == 1) ? launch_outlier_variant<3, bf16io_t>(in, out, *A, block, (hipStream_t)stream)
                                     : launch_outlier_variant<4, bf16io_t>(in, out, *A, block, (hipStream_t)stream);
    else ok = (mode == 1) ? launch_outlier_variant<3>(in, out, *A, block, (hipStream_t)stream)
                          : launch_outlier_variant<4>(in, out, *A, block, (hipStream_t)stream);
    if (!ok) { msq_set_error_("msq_outlier_fakequant: block size must be 8, 16, 32, 64 or 128"); return MSQ_ERR_UNSUPPORTED; }
    return MSQ_OK;
}
