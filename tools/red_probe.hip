// isolates the phases of msm_bucket_reduce on arbitrary (non-identity) XYZZ data
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../0g-halo2_amd/csrc/curve.h"
using namespace zg;
#define CK(x) do { hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1;} } while(0)

__device__ __forceinline__ XYZZ ldx(const XYZZ* p){ XYZZ r; const uint4* q=(const uint4*)p; uint4 v[8]; for(int i=0;i<8;i++) v[i]=q[i];
  Fe* f=(Fe*)&r; for(int i=0;i<4;i++){ f[i].l[0]=v[2*i].x;f[i].l[1]=v[2*i].y;f[i].l[2]=v[2*i].z;f[i].l[3]=v[2*i].w;f[i].l[4]=v[2*i+1].x;f[i].l[5]=v[2*i+1].y;f[i].l[6]=v[2*i+1].z;f[i].l[7]=v[2*i+1].w;} return r; }

template<int MODE>
__global__ __launch_bounds__(256) void k(const XYZZ* in, XYZZ* out, int reps) {
    __shared__ XYZZ sh[256];
    const uint32_t tid = threadIdx.x;
    XYZZ acc = ldx(in + ((blockIdx.x * 256 + tid) & 4095));
    if (MODE == 0) {  // serial merge from global
        for (int r = 0; r < reps; r++) acc = xyzz_add(acc, ldx(in + ((blockIdx.x * 256 + tid + r + 1) & 4095)));
        sh[tid] = acc;
    }
    if (MODE == 1) {  // HS scan in LDS, register temp
        sh[tid] = acc; __syncthreads();
        for (int r = 0; r < reps; r++) {
            uint32_t o = 1u << (r & 7);
            XYZZ v = xyzz_identity(); bool has = tid + o < 256;
            if (has) v = sh[tid + o];
            __syncthreads();
            if (has) sh[tid] = xyzz_add(sh[tid], v);
            __syncthreads();
        }
    }
    if (MODE == 2) {  // tree
        sh[tid] = acc; __syncthreads();
        for (int r = 0; r < reps; r++) {
            uint32_t o = 128u >> (r & 7);
            if (tid < o) sh[tid] = xyzz_add(sh[tid], sh[tid + o]);
            __syncthreads();
        }
    }
    if (MODE == 3) {  // scan, operands copied to registers first
        sh[tid] = acc; __syncthreads();
        for (int r = 0; r < reps; r++) {
            uint32_t o = 1u << (r & 7);
            bool has = tid + o < 256;
            XYZZ a = sh[tid], v = has ? sh[tid + o] : xyzz_identity();
            __syncthreads();
            if (has) a = xyzz_add(a, v);
            sh[tid] = a;
            __syncthreads();
        }
    }
    __syncthreads();
    out[blockIdx.x * 256 + tid] = sh[tid];
}

int main() {
    XYZZ *in, *out; CK(hipMalloc(&in, 4096 * sizeof(XYZZ))); CK(hipMalloc(&out, 64 * 256 * sizeof(XYZZ)));  // max grid = 64 blocks x 256 lanes
    XYZZ* h = new XYZZ[4096];
    uint32_t s = 12345; auto rnd=[&](){ s = s*1664525u+1013904223u; return s; };
    for (int i = 0; i < 4096; i++) { uint32_t* w=(uint32_t*)&h[i]; for (int j=0;j<32;j++) w[j]=rnd(); for(int f=0;f<4;f++) w[8*f+7] &= 0x0fffffff; }
    CK(hipMemcpy(in, h, 4096*sizeof(XYZZ), hipMemcpyHostToDevice));
    const char* names[4] = {"merge(global)", "scan(regtemp)", "tree", "scan(copy)"};
    for (int mode = 0; mode < 4; mode++) for (int blocks : {1, 16, 64}) {
        hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        auto launch=[&](int reps){ if(mode==0) hipLaunchKernelGGL(k<0>,dim3(blocks),dim3(256),0,0,in,out,reps);
          if(mode==1) hipLaunchKernelGGL(k<1>,dim3(blocks),dim3(256),0,0,in,out,reps);
          if(mode==2) hipLaunchKernelGGL(k<2>,dim3(blocks),dim3(256),0,0,in,out,reps);
          if(mode==3) hipLaunchKernelGGL(k<3>,dim3(blocks),dim3(256),0,0,in,out,reps); };
        launch(8); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); launch(16); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms,e0,e1));
        printf("%-14s blocks=%2d : %8.1f us for 16 steps = %6.1f us/step\n", names[mode], blocks, ms*1e3, ms*1e3/16);
    }
    return 0;
}
