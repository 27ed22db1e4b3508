// batch_src.h -- where a training call takes its rows from when the dataset is resident in HBM (osd_train_batch_source).
#pragma once
#include <stdint.h>

namespace osd {

// rows of a batch taken from a device-resident dataset, optionally mixed up (osd_train_batch_source)
struct BatchSrc {
  const float* data; int64_t ldd;      // dataset rows [N][ldd]
  const float* cond; int64_t ldc;      // condition rows [N][ldc]
  const int64_t* idx_a;                // [n] dataset row of batch row i, or null: row i
  const int64_t* idx_b;                // [n] second row of the mixup pair, or null: no mixup
  float lam, oml;                      // lam and (1 - lam), each rounded to fp32 as torch does with python scalars
};

// buffers a training call zeroes before anything accumulates into them (atomically summed gradients, the loss word)
struct ZeroList { float* ptr[128]; int64_t count[128]; int n; };

}  // namespace osd
