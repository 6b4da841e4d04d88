// Prints std::mt19937 + std::uniform_real_distribution<float>(-1,1) draws (libstdc++) so the
// oracle's restatement of the USE_ORIG random source (Compute.cpp:679-681) can be pinned to it.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
int main(int argc, char **argv)
{
    const unsigned seed = argc > 1 ? std::strtoul(argv[1], nullptr, 10) : 1u;
    const int count = argc > 2 ? std::atoi(argv[2]) : 16;
    std::mt19937 gen(seed);
    std::uniform_real_distribution<float> dist(-1.0f, 1.0f);
    for (int i = 0; i < count; i++) {
        float v = dist(gen);
        unsigned bits;
        std::memcpy(&bits, &v, 4);
        std::printf("%08x\n", bits);
    }
    return 0;
}
