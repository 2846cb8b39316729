// Regenerates the *inputs* of SURVEY.md Appendix D's whole-path golden vector
// (std::mt19937_64(20260128); draw order ao[96*5]=0.4*N, grad[3*96*5]=0.3*N,
// w[96]=0.05*U, C[5*2]=0.7*N; dm = 2 C C^T; libstdc++ 11 distributions) and
// writes them as raw little-endian doubles.  Only data is emitted; the expected
// outputs live in appendix_d.json (transcribed from SURVEY.md Appendix D, i.e.
// outputs of the reference's own device arithmetic run on the host at survey
// time).  Usage: g++ -O2 gen_appendix_d_inputs.cpp -o gen && ./gen out.bin
#include <cstdio>
#include <random>
#include <vector>
int main(int argc, char **argv)
{
    const int ng = 96, nao = 5, nocc = 2;
    std::mt19937_64 rng(20260128);
    std::normal_distribution<double> N(0.0, 1.0);
    std::uniform_real_distribution<double> U(0.0, 1.0);
    std::vector<double> ao(ng * nao), gr(3 * ng * nao), w(ng), C(nao * nocc), dm(nao * nao);
    for (auto &x : ao) x = 0.4 * N(rng);
    for (auto &x : gr) x = 0.3 * N(rng);
    for (auto &x : w) x = 0.05 * U(rng);
    for (auto &x : C) x = 0.7 * N(rng);
    for (int i = 0; i < nao; i++)
        for (int j = 0; j < nao; j++) {
            double s = 0;
            for (int k = 0; k < nocc; k++) s += C[i * nocc + k] * C[j * nocc + k];
            dm[i * nao + j] = 2 * s;
        }
    FILE *f = fopen(argc > 1 ? argv[1] : "appendix_d_inputs.bin", "wb");
    if (!f) return 1;
    fwrite(ao.data(), 8, ao.size(), f);
    fwrite(gr.data(), 8, gr.size(), f);
    fwrite(w.data(), 8, w.size(), f);
    fwrite(dm.data(), 8, dm.size(), f);
    fclose(f);
    return 0;
}
