// Compile-only (g++ -c): instantiates every template of include/UnifiedCvo/pcl_interop.hpp with the mock
// pcl::PointCloud of tests/mock_include and the 192-byte cvo::CvoPoint record, so that the adaptor header has been through
// a compiler (the overloads it forwards to live in libcvo_gpu_img_lib; their behaviour is tests/test_cpp_host.py's).
#include "pcl_interop.hpp"

#ifndef UNIFIEDCVO_HAS_PCL
#error "the mock pcl/point_cloud.h was not found: pcl_interop.hpp compiled to nothing"
#endif

int interop_pcl_instantiate(const cvo::CvoGPU& cvo, const pcl::PointCloud<cvo::CvoPoint>& a,
                            const pcl::PointCloud<cvo::CvoPoint>& b, float* out) {
  cvo::Mat4f T = cvo::Mat4f::Identity(), R = cvo::Mat4f::Identity();
  cvo::Association assoc;
  double seconds = 0;
  const int rc = cvo::align(cvo, a, b, T, R, &assoc, &seconds);
  out[0] = cvo::inner_product_gpu(cvo, a, b, T, 0.3f);
  out[1] = cvo::function_angle(cvo, a, b, T, 0.3f, true);
  out[2] = cvo::function_angle(cvo, a, b, T, 0.3f);
  return rc + cvo::align(cvo, a, b, T, R);
}
