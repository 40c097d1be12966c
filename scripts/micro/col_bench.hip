// Developer micro-benchmark of k_lu_col in isolation (includes lu.hip to reach the static kernel).
#include "../../runmat_amd/csrc/lu.hip"
namespace rmhip {
int fail(int code, const char*, ...) { return code; }
void set_error(const char*, ...) {}
int launch_dgemm(Context*, size_t, size_t, size_t, double, const double*, size_t, const double*, size_t, double, double*, size_t) { return 0; }
}
using namespace rmhip;
int main(int argc, char** argv) {
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    for (size_t rows : {2048ul, 8192ul, 16384ul}) {
        const size_t lda = rows + 32, cols = 64;
        double* A; hipMalloc(&A, lda * cols * 8);
        std::vector<double> h(lda * cols);
        for (size_t i = 0; i < h.size(); ++i) h[i] = (double)((i * 2654435761u) % 1000) / 1000.0 + 0.001;
        hipMemcpy(A, h.data(), h.size() * 8, hipMemcpyHostToDevice);
        char* blk; hipMalloc(&blk, 1 << 20); hipMemset(blk, 0, 1 << 20);
        int* ipiv = (int*)blk; int* info = ipiv + rows;
        int* pos_of = (int*)(blk + 100000); int* row_at = (int*)(blk + 200000); int* prow = (int*)(blk + 600000); double* ca = (double*)(blk + 300000); int* cp = (int*)(blk + 400000); int* cr = (int*)(blk + 500000);
        const int nb = (int)((rows + 63) / 64);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int dbg : {0}) {
            hipMemcpy(A, h.data(), h.size() * 8, hipMemcpyHostToDevice);
            const int reps = 20;
            hipEventRecord(e0, st);
            for (int rep = 0; rep < reps; ++rep) {
                hipLaunchKernelGGL(k_lu_col, dim3(nb), dim3(PANEL_THREADS), 0, st, A, lda, rows, 0, -1, 64, 1, nb, pos_of, row_at, prow, ipiv, info, ca, cp, cr);
                for (int k = 0; k < 64; ++k)
                    hipLaunchKernelGGL(k_lu_col, dim3(nb), dim3(PANEL_THREADS), 0, st, A, lda, rows, 0, k, 64, 0, nb, pos_of, row_at, prow, ipiv, info, ca, cp, cr);
            }
            hipEventRecord(e1, st); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("rows=%zu dbg=%d: %.2f us per column launch\n", rows, dbg, ms * 1000.f / (reps * 65));
        }
        hipFree(A); hipFree(blk);
    }
    return 0;
}
