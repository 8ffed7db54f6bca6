// orbx_math.h -- scalar float/double helpers shared by host and gfx950 device code.
//
// Everything here must evaluate bit-identically on the host and on the GPU, so: no libm
// transcendental calls, no FMA contraction (the library is built with -ffp-contract=off and the
// float expressions below are written as separate operations), IEEE division.
#pragma once

#include <stdint.h>

#if defined(__HIPCC__)
#define ORBX_HD __host__ __device__ __forceinline__
#else
#define ORBX_HD inline
#endif

namespace orbx {

// cvRound (OpenCV core/fast_math.hpp): round-half-to-even for float arguments.
ORBX_HD int cv_round_f(float v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __float2int_rn(v);
#else
    return (int)__builtin_nearbyintf(v);
#endif
}

// cv::fastAtan2(y, x) in degrees [0, 360) -- the polynomial of OpenCV's atan_f32 (mathfuncs_core),
// as used by IC_Angle (reference src/ORBextractor.cc:102).  f32, no FMA.
ORBX_HD float fast_atan2_deg(float y, float x)
{
    const float scale = (float)(180.0 / 3.141592653589793238462643383279502884);
    const float p1 = 0.9997878412794807f * scale;
    const float p3 = -0.3258083974640975f * scale;
    const float p5 = 0.1555786518463281f * scale;
    const float p7 = -0.04432655554792128f * scale;
    const float eps = 2.2204460492503131e-16f;     // (float)DBL_EPSILON
    const float ax = __builtin_fabsf(x), ay = __builtin_fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + eps);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + eps);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

// cos/sin of a float angle (radians), evaluated in double by a fixed algorithm and rounded to float
// (reference src/ORBextractor.cc:111-112 calls libm on a float).  Cody-Waite reduction by pi/2 and
// Taylor polynomials; |error| ~1e-16 before the final rounding.  Valid for |x| < ~1e5.
ORBX_HD void sincos_f32(float angle, float* c_out, float* s_out)
{
    const double x = (double)angle;
    const double kd = __builtin_rint(x * 0.63661977236758134308);
    const int k = (int)kd;
    const double r = (x - kd * 1.57079632679489655800e+00) - kd * 6.12323399573676603587e-17;
    const double z = r * r;
    double sp = 2.81145725434552075980e-15;
    sp = sp * z + -7.64716373181981647590e-13;
    sp = sp * z + 1.60590438368216145994e-10;
    sp = sp * z + -2.50521083854417187751e-08;
    sp = sp * z + 2.75573192239858906526e-06;
    sp = sp * z + -1.98412698412698412698e-04;
    sp = sp * z + 8.33333333333333333333e-03;
    sp = sp * z + -1.66666666666666666667e-01;
    const double s = r + r * (sp * z);
    double cp = -1.56192069685862264433e-16;
    cp = cp * z + 4.77947733238738529744e-14;
    cp = cp * z + -1.14707455977297247139e-11;
    cp = cp * z + 2.08767569878680989792e-09;
    cp = cp * z + -2.75573192239858906526e-07;
    cp = cp * z + 2.48015873015873015873e-05;
    cp = cp * z + -1.38888888888888888889e-03;
    cp = cp * z + 4.16666666666666666667e-02;
    cp = cp * z + -5.00000000000000000000e-01;
    const double c = 1.0 + cp * z;
    double cc, ss;
    switch (k & 3) {
        case 0: cc = c; ss = s; break;
        case 1: cc = -s; ss = c; break;
        case 2: cc = -c; ss = -s; break;
        default: cc = s; ss = -c; break;
    }
    *c_out = (float)cc;
    *s_out = (float)ss;
}

}  // namespace orbx
