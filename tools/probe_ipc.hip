// Feasibility probe for the direct per-step exchange on a one-GPU box: two PROCESSES share a device
// mailbox through hipIpc; a kernel of the child stores data + flag with system scope, a kernel of the parent polls the
// flag (bounded) and reads the data.  Prints what worked.  build: hipcc --offload-arch=gfx950 -O2 -o _probe_ipc probe_ipc.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sys/wait.h>
#include <unistd.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

__global__ void writer(double* box, unsigned long long* flag, unsigned long long epoch)
{
    if (threadIdx.x < 64) __hip_atomic_store(&box[threadIdx.x], 1000.0 * (double)epoch + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ void reader(const double* box, const unsigned long long* flag, unsigned long long epoch, double* out, int* status)
{
    __shared__ int ok;
    if (threadIdx.x == 0) {
        const unsigned long long t0 = wall_clock64();  // 100 MHz
        ok = 0;
        while (wall_clock64() - t0 < 200000000ull) {   // 2 s bound
            if (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) == epoch) { ok = 1; break; }
            __builtin_amdgcn_s_sleep(20);
        }
    }
    __syncthreads();
    if (threadIdx.x < 64) out[threadIdx.x] = ok ? __hip_atomic_load(&box[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : -1.0;
    if (threadIdx.x == 0) *status = ok;
}

int main()
{
    int to_child[2], to_parent[2];
    if (pipe(to_child) || pipe(to_parent)) return 1;
    const pid_t pid = fork();  // before any HIP call
    if (pid == 0) {            // child: opens the parent's mailbox and writes into it
        hipIpcMemHandle_t h;
        if (read(to_child[0], &h, sizeof h) != (ssize_t)sizeof h) return 3;
        void* p = nullptr;
        CK(hipSetDevice(0));
        CK(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
        double* box = (double*)p;
        unsigned long long* flag = (unsigned long long*)(box + 64);
        for (unsigned long long e = 1; e <= 3; ++e) {
            char go;
            if (read(to_child[0], &go, 1) != 1) return 3;
            hipLaunchKernelGGL(writer, dim3(1), dim3(64), 0, 0, box, flag, e);
            CK(hipDeviceSynchronize());
        }
        CK(hipIpcCloseMemHandle(p));
        char done = 1;
        (void)!write(to_parent[1], &done, 1);
        return 0;
    }
    CK(hipSetDevice(0));
    for (int fine = 1; fine >= 0; --fine) {
        void* p = nullptr;
        hipError_t e = fine ? hipExtMallocWithFlags(&p, 4096, hipDeviceMallocFinegrained) : hipMalloc(&p, 4096);
        printf("alloc %s: %s\n", fine ? "fine-grained" : "plain", hipGetErrorString(e));
        if (e != hipSuccess) continue;
        CK(hipMemset(p, 0, 4096));
        hipIpcMemHandle_t h;
        e = hipIpcGetMemHandle(&h, p);
        printf("hipIpcGetMemHandle: %s\n", hipGetErrorString(e));
        if (e != hipSuccess) { (void)hipFree(p); continue; }
        (void)!write(to_child[1], &h, sizeof h);
        double* box = (double*)p;
        unsigned long long* flag = (unsigned long long*)(box + 64);
        double* out; int* status;
        CK(hipHostMalloc((void**)&out, 64 * 8)); CK(hipHostMalloc((void**)&status, 4));
        for (unsigned long long ep = 1; ep <= 3; ++ep) {
            hipLaunchKernelGGL(reader, dim3(1), dim3(64), 0, 0, box, flag, ep, out, status);  // polls while the child writes
            char go = 1;
            (void)!write(to_child[1], &go, 1);
            CK(hipDeviceSynchronize());
            printf("epoch %llu: status %d, data[0] %.1f data[63] %.1f (expect %.1f %.1f)\n", ep, *status, out[0], out[63], 1000.0 * ep, 1000.0 * ep + 63);
        }
        char done;
        (void)!read(to_parent[0], &done, 1);
        int st = 0;
        waitpid(pid, &st, 0);
        printf("child exit %d\n", WEXITSTATUS(st));
        return 0;
    }
    return 1;
}
