// Exposes the REFERENCE's own morton.hpp through a C ABI so tests can pin the oracle's and the
// product's Morton codec against it.  The header is compiled where it lies (path given by
// -DBLOK_REF_MORTON_HPP=...); nothing from the reference is copied into this repository, and the
// resulting oracle/_ref/libref_morton.so is git-ignored.  morton.hpp is the only file on the trace
// path that compiles without glm / Vulkan headers (see blok_oracle.cpp, PARITY PIN STATUS).
#include BLOK_REF_MORTON_HPP

extern "C" {
unsigned long long ref_morton_encode(int x, int y, int z) { return blok::morton3d::encode(x, y, z); }
void ref_morton_decode(unsigned long long code, int* x, int* y, int* z) {
    int32_t a, b, c;
    blok::morton3d::decode(code, a, b, c);
    *x = a; *y = b; *z = c;
}
unsigned ref_morton_octant(unsigned long long code, unsigned maxDepth, unsigned level) {
    return blok::morton3d::octantFromCode(code, maxDepth, level);
}
}
