// Times rocSOLVER's symmetric eigensolvers at the SCF sizes: syevd (what torch.linalg.eigh calls through
// hipSOLVER), syevj (Jacobi).  hipcc tools/eigh_probe.cpp -o tools/eigh_probe -lrocsolver -lrocblas
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <rocsolver/rocsolver.h>
#include <chrono>
#include <cstdio>
#include <random>
#include <vector>

int main()
{
    rocblas_handle h;
    rocblas_create_handle(&h);
    for (int n : {114, 246, 494, 1150}) {
        std::vector<double> A((size_t)n * n);
        std::mt19937_64 rng(n);
        std::normal_distribution<double> nd;
        for (int i = 0; i < n; ++i)
            for (int j = 0; j <= i; ++j) A[(size_t)i * n + j] = A[(size_t)j * n + i] = nd(rng);
        double *dA, *dW, *dE, *dres;
        rocblas_int *dinfo, *dsweeps;
        hipMalloc(&dA, sizeof(double) * n * n); hipMalloc(&dW, sizeof(double) * n); hipMalloc(&dE, sizeof(double) * n);
        hipMalloc(&dres, sizeof(double)); hipMalloc(&dinfo, sizeof(rocblas_int)); hipMalloc(&dsweeps, sizeof(rocblas_int));
        auto run = [&](int which) {
            double best = 1e30;
            for (int rep = 0; rep < 4; ++rep) {
                hipMemcpy(dA, A.data(), sizeof(double) * n * n, hipMemcpyHostToDevice);
                hipDeviceSynchronize();
                auto t0 = std::chrono::steady_clock::now();
                if (which == 0) rocsolver_dsyevd(h, rocblas_evect_original, rocblas_fill_lower, n, dA, n, dW, dE, dinfo);
                else rocsolver_dsyevj(h, rocblas_esort_ascending, rocblas_evect_original, rocblas_fill_lower, n, dA, n, 1e-14, dres, 30, dsweeps, dW, dinfo);
                hipDeviceSynchronize();
                double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
                if (ms < best) best = ms;
            }
            return best;
        };
        double t_d = run(0), t_j = run(1);
        rocblas_int sw = 0; hipMemcpy(&sw, dsweeps, sizeof sw, hipMemcpyDeviceToHost);
        printf("n=%5d  syevd %8.2f ms   syevj %8.2f ms (%d sweeps)\n", n, t_d, t_j, sw);
        hipFree(dA); hipFree(dW); hipFree(dE); hipFree(dres); hipFree(dinfo); hipFree(dsweeps);
    }
    rocblas_destroy_handle(h);
    return 0;
}
