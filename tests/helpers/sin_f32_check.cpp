// CPU check of mixlab_amd/csrc/mx_sin_f32.hpp (the same header the device compiles): reads f64 arguments (raw, little-endian) from stdin and writes, per
// argument, three floats: the slow path's result (double-double, correctly rounded), the Ziv result over the host libm's sin, and (float)sin(x) of the host libm.
//   g++ -O2 -ffp-contract=off -o sin_f32_check sin_f32_check.cpp
#include <cstdio>
#include <vector>

#include "../../mixlab_amd/csrc/mx_sin_f32.hpp"

int main() {
    std::vector<double> xs;
    double x;
    while (fread(&x, sizeof x, 1, stdin) == 1) xs.push_back(x);
    for (double v : xs) {
        const float out[3] = {mx::dd_to_f32(mx::sin_dd(v)), mx::sin_f32_from(v, sin(v), 4.6), (float)sin(v)};
        fwrite(out, sizeof(float), 3, stdout);
    }
    return 0;
}
