
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 4096
__global__ void k_v_pk_add_u16(unsigned* out, unsigned seed) {
  unsigned a[16]; for (int i=0;i<16;++i) a[i]=threadIdx.x*17+i+seed; unsigned b=seed|1;
  for (int it=0; it<ITER; ++it) asm volatile("v_pk_add_u16 %0, %0, %16\nv_pk_add_u16 %1, %1, %16\nv_pk_add_u16 %2, %2, %16\nv_pk_add_u16 %3, %3, %16\nv_pk_add_u16 %4, %4, %16\nv_pk_add_u16 %5, %5, %16\nv_pk_add_u16 %6, %6, %16\nv_pk_add_u16 %7, %7, %16\nv_pk_add_u16 %8, %8, %16\nv_pk_add_u16 %9, %9, %16\nv_pk_add_u16 %10, %10, %16\nv_pk_add_u16 %11, %11, %16\nv_pk_add_u16 %12, %12, %16\nv_pk_add_u16 %13, %13, %16\nv_pk_add_u16 %14, %14, %16\nv_pk_add_u16 %15, %15, %16\n" : "+v"(a[0]),"+v"(a[1]),"+v"(a[2]),"+v"(a[3]),"+v"(a[4]),"+v"(a[5]),"+v"(a[6]),"+v"(a[7]),"+v"(a[8]),"+v"(a[9]),"+v"(a[10]),"+v"(a[11]),"+v"(a[12]),"+v"(a[13]),"+v"(a[14]),"+v"(a[15]) : "v"(b));
  unsigned s=0; for (int i=0;i<16;++i) s^=a[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }
__global__ void k_v_pk_max_i16(unsigned* out, unsigned seed) {
  unsigned a[16]; for (int i=0;i<16;++i) a[i]=threadIdx.x*17+i+seed; unsigned b=seed|1;
  for (int it=0; it<ITER; ++it) asm volatile("v_pk_max_i16 %0, %0, %16\nv_pk_max_i16 %1, %1, %16\nv_pk_max_i16 %2, %2, %16\nv_pk_max_i16 %3, %3, %16\nv_pk_max_i16 %4, %4, %16\nv_pk_max_i16 %5, %5, %16\nv_pk_max_i16 %6, %6, %16\nv_pk_max_i16 %7, %7, %16\nv_pk_max_i16 %8, %8, %16\nv_pk_max_i16 %9, %9, %16\nv_pk_max_i16 %10, %10, %16\nv_pk_max_i16 %11, %11, %16\nv_pk_max_i16 %12, %12, %16\nv_pk_max_i16 %13, %13, %16\nv_pk_max_i16 %14, %14, %16\nv_pk_max_i16 %15, %15, %16\n" : "+v"(a[0]),"+v"(a[1]),"+v"(a[2]),"+v"(a[3]),"+v"(a[4]),"+v"(a[5]),"+v"(a[6]),"+v"(a[7]),"+v"(a[8]),"+v"(a[9]),"+v"(a[10]),"+v"(a[11]),"+v"(a[12]),"+v"(a[13]),"+v"(a[14]),"+v"(a[15]) : "v"(b));
  unsigned s=0; for (int i=0;i<16;++i) s^=a[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }
__global__ void k_v_pk_max_u16(unsigned* out, unsigned seed) {
  unsigned a[16]; for (int i=0;i<16;++i) a[i]=threadIdx.x*17+i+seed; unsigned b=seed|1;
  for (int it=0; it<ITER; ++it) asm volatile("v_pk_max_u16 %0, %0, %16\nv_pk_max_u16 %1, %1, %16\nv_pk_max_u16 %2, %2, %16\nv_pk_max_u16 %3, %3, %16\nv_pk_max_u16 %4, %4, %16\nv_pk_max_u16 %5, %5, %16\nv_pk_max_u16 %6, %6, %16\nv_pk_max_u16 %7, %7, %16\nv_pk_max_u16 %8, %8, %16\nv_pk_max_u16 %9, %9, %16\nv_pk_max_u16 %10, %10, %16\nv_pk_max_u16 %11, %11, %16\nv_pk_max_u16 %12, %12, %16\nv_pk_max_u16 %13, %13, %16\nv_pk_max_u16 %14, %14, %16\nv_pk_max_u16 %15, %15, %16\n" : "+v"(a[0]),"+v"(a[1]),"+v"(a[2]),"+v"(a[3]),"+v"(a[4]),"+v"(a[5]),"+v"(a[6]),"+v"(a[7]),"+v"(a[8]),"+v"(a[9]),"+v"(a[10]),"+v"(a[11]),"+v"(a[12]),"+v"(a[13]),"+v"(a[14]),"+v"(a[15]) : "v"(b));
  unsigned s=0; for (int i=0;i<16;++i) s^=a[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }
__global__ void k_v_pk_min_u16(unsigned* out, unsigned seed) {
  unsigned a[16]; for (int i=0;i<16;++i) a[i]=threadIdx.x*17+i+seed; unsigned b=seed|1;
  for (int it=0; it<ITER; ++it) asm volatile("v_pk_min_u16 %0, %0, %16\nv_pk_min_u16 %1, %1, %16\nv_pk_min_u16 %2, %2, %16\nv_pk_min_u16 %3, %3, %16\nv_pk_min_u16 %4, %4, %16\nv_pk_min_u16 %5, %5, %16\nv_pk_min_u16 %6, %6, %16\nv_pk_min_u16 %7, %7, %16\nv_pk_min_u16 %8, %8, %16\nv_pk_min_u16 %9, %9, %16\nv_pk_min_u16 %10, %10, %16\nv_pk_min_u16 %11, %11, %16\nv_pk_min_u16 %12, %12, %16\nv_pk_min_u16 %13, %13, %16\nv_pk_min_u16 %14, %14, %16\nv_pk_min_u16 %15, %15, %16\n" : "+v"(a[0]),"+v"(a[1]),"+v"(a[2]),"+v"(a[3]),"+v"(a[4]),"+v"(a[5]),"+v"(a[6]),"+v"(a[7]),"+v"(a[8]),"+v"(a[9]),"+v"(a[10]),"+v"(a[11]),"+v"(a[12]),"+v"(a[13]),"+v"(a[14]),"+v"(a[15]) : "v"(b));
  unsigned s=0; for (int i=0;i<16;++i) s^=a[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }
__global__ void k_v_add_u32(unsigned* out, unsigned seed) {
  unsigned a[16]; for (int i=0;i<16;++i) a[i]=threadIdx.x*17+i+seed; unsigned b=seed|1;
  for (int it=0; it<ITER; ++it) asm volatile("v_add_u32 %0, %0, %16\nv_add_u32 %1, %1, %16\nv_add_u32 %2, %2, %16\nv_add_u32 %3, %3, %16\nv_add_u32 %4, %4, %16\nv_add_u32 %5, %5, %16\nv_add_u32 %6, %6, %16\nv_add_u32 %7, %7, %16\nv_add_u32 %8, %8, %16\nv_add_u32 %9, %9, %16\nv_add_u32 %10, %10, %16\nv_add_u32 %11, %11, %16\nv_add_u32 %12, %12, %16\nv_add_u32 %13, %13, %16\nv_add_u32 %14, %14, %16\nv_add_u32 %15, %15, %16\n" : "+v"(a[0]),"+v"(a[1]),"+v"(a[2]),"+v"(a[3]),"+v"(a[4]),"+v"(a[5]),"+v"(a[6]),"+v"(a[7]),"+v"(a[8]),"+v"(a[9]),"+v"(a[10]),"+v"(a[11]),"+v"(a[12]),"+v"(a[13]),"+v"(a[14]),"+v"(a[15]) : "v"(b));
  unsigned s=0; for (int i=0;i<16;++i) s^=a[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }
__global__ void k_v_sub_u32(unsigned* out, unsigned seed) {
  unsigned a[16]; for (int i=0;i<16;++i) a[i]=threadIdx.x*17+i+seed; unsigned b=seed|1;
  for (int it=0; it<ITER; ++it) asm volatile("v_sub_u32 %0, %0, %16\nv_sub_u32 %1, %1, %16\nv_sub_u32 %2, %2, %16\nv_sub_u32 %3, %3, %16\nv_sub_u32 %4, %4, %16\nv_sub_u32 %5, %5, %16\nv_sub_u32 %6, %6, %16\nv_sub_u32 %7, %7, %16\nv_sub_u32 %8, %8, %16\nv_sub_u32 %9, %9, %16\nv_sub_u32 %10, %10, %16\nv_sub_u32 %11, %11, %16\nv_sub_u32 %12, %12, %16\nv_sub_u32 %13, %13, %16\nv_sub_u32 %14, %14, %16\nv_sub_u32 %15, %15, %16\n" : "+v"(a[0]),"+v"(a[1]),"+v"(a[2]),"+v"(a[3]),"+v"(a[4]),"+v"(a[5]),"+v"(a[6]),"+v"(a[7]),"+v"(a[8]),"+v"(a[9]),"+v"(a[10]),"+v"(a[11]),"+v"(a[12]),"+v"(a[13]),"+v"(a[14]),"+v"(a[15]) : "v"(b));
  unsigned s=0; for (int i=0;i<16;++i) s^=a[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }
__global__ void k_v_max_i32(unsigned* out, unsigned seed) {
  unsigned a[16]; for (int i=0;i<16;++i) a[i]=threadIdx.x*17+i+seed; unsigned b=seed|1;
  for (int it=0; it<ITER; ++it) asm volatile("v_max_i32 %0, %0, %16\nv_max_i32 %1, %1, %16\nv_max_i32 %2, %2, %16\nv_max_i32 %3, %3, %16\nv_max_i32 %4, %4, %16\nv_max_i32 %5, %5, %16\nv_max_i32 %6, %6, %16\nv_max_i32 %7, %7, %16\nv_max_i32 %8, %8, %16\nv_max_i32 %9, %9, %16\nv_max_i32 %10, %10, %16\nv_max_i32 %11, %11, %16\nv_max_i32 %12, %12, %16\nv_max_i32 %13, %13, %16\nv_max_i32 %14, %14, %16\nv_max_i32 %15, %15, %16\n" : "+v"(a[0]),"+v"(a[1]),"+v"(a[2]),"+v"(a[3]),"+v"(a[4]),"+v"(a[5]),"+v"(a[6]),"+v"(a[7]),"+v"(a[8]),"+v"(a[9]),"+v"(a[10]),"+v"(a[11]),"+v"(a[12]),"+v"(a[13]),"+v"(a[14]),"+v"(a[15]) : "v"(b));
  unsigned s=0; for (int i=0;i<16;++i) s^=a[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }
__global__ void k_v_max_u32(unsigned* out, unsigned seed) {
  unsigned a[16]; for (int i=0;i<16;++i) a[i]=threadIdx.x*17+i+seed; unsigned b=seed|1;
  for (int it=0; it<ITER; ++it) asm volatile("v_max_u32 %0, %0, %16\nv_max_u32 %1, %1, %16\nv_max_u32 %2, %2, %16\nv_max_u32 %3, %3, %16\nv_max_u32 %4, %4, %16\nv_max_u32 %5, %5, %16\nv_max_u32 %6, %6, %16\nv_max_u32 %7, %7, %16\nv_max_u32 %8, %8, %16\nv_max_u32 %9, %9, %16\nv_max_u32 %10, %10, %16\nv_max_u32 %11, %11, %16\nv_max_u32 %12, %12, %16\nv_max_u32 %13, %13, %16\nv_max_u32 %14, %14, %16\nv_max_u32 %15, %15, %16\n" : "+v"(a[0]),"+v"(a[1]),"+v"(a[2]),"+v"(a[3]),"+v"(a[4]),"+v"(a[5]),"+v"(a[6]),"+v"(a[7]),"+v"(a[8]),"+v"(a[9]),"+v"(a[10]),"+v"(a[11]),"+v"(a[12]),"+v"(a[13]),"+v"(a[14]),"+v"(a[15]) : "v"(b));
  unsigned s=0; for (int i=0;i<16;++i) s^=a[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }
__global__ void k_v_min_u32(unsigned* out, unsigned seed) {
  unsigned a[16]; for (int i=0;i<16;++i) a[i]=threadIdx.x*17+i+seed; unsigned b=seed|1;
  for (int it=0; it<ITER; ++it) asm volatile("v_min_u32 %0, %0, %16\nv_min_u32 %1, %1, %16\nv_min_u32 %2, %2, %16\nv_min_u32 %3, %3, %16\nv_min_u32 %4, %4, %16\nv_min_u32 %5, %5, %16\nv_min_u32 %6, %6, %16\nv_min_u32 %7, %7, %16\nv_min_u32 %8, %8, %16\nv_min_u32 %9, %9, %16\nv_min_u32 %10, %10, %16\nv_min_u32 %11, %11, %16\nv_min_u32 %12, %12, %16\nv_min_u32 %13, %13, %16\nv_min_u32 %14, %14, %16\nv_min_u32 %15, %15, %16\n" : "+v"(a[0]),"+v"(a[1]),"+v"(a[2]),"+v"(a[3]),"+v"(a[4]),"+v"(a[5]),"+v"(a[6]),"+v"(a[7]),"+v"(a[8]),"+v"(a[9]),"+v"(a[10]),"+v"(a[11]),"+v"(a[12]),"+v"(a[13]),"+v"(a[14]),"+v"(a[15]) : "v"(b));
  unsigned s=0; for (int i=0;i<16;++i) s^=a[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }
__global__ void k_v_xor_b32(unsigned* out, unsigned seed) {
  unsigned a[16]; for (int i=0;i<16;++i) a[i]=threadIdx.x*17+i+seed; unsigned b=seed|1;
  for (int it=0; it<ITER; ++it) asm volatile("v_xor_b32 %0, %0, %16\nv_xor_b32 %1, %1, %16\nv_xor_b32 %2, %2, %16\nv_xor_b32 %3, %3, %16\nv_xor_b32 %4, %4, %16\nv_xor_b32 %5, %5, %16\nv_xor_b32 %6, %6, %16\nv_xor_b32 %7, %7, %16\nv_xor_b32 %8, %8, %16\nv_xor_b32 %9, %9, %16\nv_xor_b32 %10, %10, %16\nv_xor_b32 %11, %11, %16\nv_xor_b32 %12, %12, %16\nv_xor_b32 %13, %13, %16\nv_xor_b32 %14, %14, %16\nv_xor_b32 %15, %15, %16\n" : "+v"(a[0]),"+v"(a[1]),"+v"(a[2]),"+v"(a[3]),"+v"(a[4]),"+v"(a[5]),"+v"(a[6]),"+v"(a[7]),"+v"(a[8]),"+v"(a[9]),"+v"(a[10]),"+v"(a[11]),"+v"(a[12]),"+v"(a[13]),"+v"(a[14]),"+v"(a[15]) : "v"(b));
  unsigned s=0; for (int i=0;i<16;++i) s^=a[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }
__global__ void k_v_and_b32(unsigned* out, unsigned seed) {
  unsigned a[16]; for (int i=0;i<16;++i) a[i]=threadIdx.x*17+i+seed; unsigned b=seed|1;
  for (int it=0; it<ITER; ++it) asm volatile("v_and_b32 %0, %0, %16\nv_and_b32 %1, %1, %16\nv_and_b32 %2, %2, %16\nv_and_b32 %3, %3, %16\nv_and_b32 %4, %4, %16\nv_and_b32 %5, %5, %16\nv_and_b32 %6, %6, %16\nv_and_b32 %7, %7, %16\nv_and_b32 %8, %8, %16\nv_and_b32 %9, %9, %16\nv_and_b32 %10, %10, %16\nv_and_b32 %11, %11, %16\nv_and_b32 %12, %12, %16\nv_and_b32 %13, %13, %16\nv_and_b32 %14, %14, %16\nv_and_b32 %15, %15, %16\n" : "+v"(a[0]),"+v"(a[1]),"+v"(a[2]),"+v"(a[3]),"+v"(a[4]),"+v"(a[5]),"+v"(a[6]),"+v"(a[7]),"+v"(a[8]),"+v"(a[9]),"+v"(a[10]),"+v"(a[11]),"+v"(a[12]),"+v"(a[13]),"+v"(a[14]),"+v"(a[15]) : "v"(b));
  unsigned s=0; for (int i=0;i<16;++i) s^=a[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }
__global__ void k_v_max_f32(unsigned* out, unsigned seed) {
  unsigned a[16]; for (int i=0;i<16;++i) a[i]=threadIdx.x*17+i+seed; unsigned b=seed|1;
  for (int it=0; it<ITER; ++it) asm volatile("v_max_f32 %0, %0, %16\nv_max_f32 %1, %1, %16\nv_max_f32 %2, %2, %16\nv_max_f32 %3, %3, %16\nv_max_f32 %4, %4, %16\nv_max_f32 %5, %5, %16\nv_max_f32 %6, %6, %16\nv_max_f32 %7, %7, %16\nv_max_f32 %8, %8, %16\nv_max_f32 %9, %9, %16\nv_max_f32 %10, %10, %16\nv_max_f32 %11, %11, %16\nv_max_f32 %12, %12, %16\nv_max_f32 %13, %13, %16\nv_max_f32 %14, %14, %16\nv_max_f32 %15, %15, %16\n" : "+v"(a[0]),"+v"(a[1]),"+v"(a[2]),"+v"(a[3]),"+v"(a[4]),"+v"(a[5]),"+v"(a[6]),"+v"(a[7]),"+v"(a[8]),"+v"(a[9]),"+v"(a[10]),"+v"(a[11]),"+v"(a[12]),"+v"(a[13]),"+v"(a[14]),"+v"(a[15]) : "v"(b));
  unsigned s=0; for (int i=0;i<16;++i) s^=a[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }
__global__ void k_v_add_f32(unsigned* out, unsigned seed) {
  unsigned a[16]; for (int i=0;i<16;++i) a[i]=threadIdx.x*17+i+seed; unsigned b=seed|1;
  for (int it=0; it<ITER; ++it) asm volatile("v_add_f32 %0, %0, %16\nv_add_f32 %1, %1, %16\nv_add_f32 %2, %2, %16\nv_add_f32 %3, %3, %16\nv_add_f32 %4, %4, %16\nv_add_f32 %5, %5, %16\nv_add_f32 %6, %6, %16\nv_add_f32 %7, %7, %16\nv_add_f32 %8, %8, %16\nv_add_f32 %9, %9, %16\nv_add_f32 %10, %10, %16\nv_add_f32 %11, %11, %16\nv_add_f32 %12, %12, %16\nv_add_f32 %13, %13, %16\nv_add_f32 %14, %14, %16\nv_add_f32 %15, %15, %16\n" : "+v"(a[0]),"+v"(a[1]),"+v"(a[2]),"+v"(a[3]),"+v"(a[4]),"+v"(a[5]),"+v"(a[6]),"+v"(a[7]),"+v"(a[8]),"+v"(a[9]),"+v"(a[10]),"+v"(a[11]),"+v"(a[12]),"+v"(a[13]),"+v"(a[14]),"+v"(a[15]) : "v"(b));
  unsigned s=0; for (int i=0;i<16;++i) s^=a[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }
__global__ void k_v_pk_add_f16(unsigned* out, unsigned seed) {
  unsigned a[16]; for (int i=0;i<16;++i) a[i]=threadIdx.x*17+i+seed; unsigned b=seed|1;
  for (int it=0; it<ITER; ++it) asm volatile("v_pk_add_f16 %0, %0, %16\nv_pk_add_f16 %1, %1, %16\nv_pk_add_f16 %2, %2, %16\nv_pk_add_f16 %3, %3, %16\nv_pk_add_f16 %4, %4, %16\nv_pk_add_f16 %5, %5, %16\nv_pk_add_f16 %6, %6, %16\nv_pk_add_f16 %7, %7, %16\nv_pk_add_f16 %8, %8, %16\nv_pk_add_f16 %9, %9, %16\nv_pk_add_f16 %10, %10, %16\nv_pk_add_f16 %11, %11, %16\nv_pk_add_f16 %12, %12, %16\nv_pk_add_f16 %13, %13, %16\nv_pk_add_f16 %14, %14, %16\nv_pk_add_f16 %15, %15, %16\n" : "+v"(a[0]),"+v"(a[1]),"+v"(a[2]),"+v"(a[3]),"+v"(a[4]),"+v"(a[5]),"+v"(a[6]),"+v"(a[7]),"+v"(a[8]),"+v"(a[9]),"+v"(a[10]),"+v"(a[11]),"+v"(a[12]),"+v"(a[13]),"+v"(a[14]),"+v"(a[15]) : "v"(b));
  unsigned s=0; for (int i=0;i<16;++i) s^=a[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }
__global__ void k_v_pk_max_f16(unsigned* out, unsigned seed) {
  unsigned a[16]; for (int i=0;i<16;++i) a[i]=threadIdx.x*17+i+seed; unsigned b=seed|1;
  for (int it=0; it<ITER; ++it) asm volatile("v_pk_max_f16 %0, %0, %16\nv_pk_max_f16 %1, %1, %16\nv_pk_max_f16 %2, %2, %16\nv_pk_max_f16 %3, %3, %16\nv_pk_max_f16 %4, %4, %16\nv_pk_max_f16 %5, %5, %16\nv_pk_max_f16 %6, %6, %16\nv_pk_max_f16 %7, %7, %16\nv_pk_max_f16 %8, %8, %16\nv_pk_max_f16 %9, %9, %16\nv_pk_max_f16 %10, %10, %16\nv_pk_max_f16 %11, %11, %16\nv_pk_max_f16 %12, %12, %16\nv_pk_max_f16 %13, %13, %16\nv_pk_max_f16 %14, %14, %16\nv_pk_max_f16 %15, %15, %16\n" : "+v"(a[0]),"+v"(a[1]),"+v"(a[2]),"+v"(a[3]),"+v"(a[4]),"+v"(a[5]),"+v"(a[6]),"+v"(a[7]),"+v"(a[8]),"+v"(a[9]),"+v"(a[10]),"+v"(a[11]),"+v"(a[12]),"+v"(a[13]),"+v"(a[14]),"+v"(a[15]) : "v"(b));
  unsigned s=0; for (int i=0;i<16;++i) s^=a[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }
__global__ void k_v_max_u16(unsigned* out, unsigned seed) {
  unsigned a[16]; for (int i=0;i<16;++i) a[i]=threadIdx.x*17+i+seed; unsigned b=seed|1;
  for (int it=0; it<ITER; ++it) asm volatile("v_max_u16 %0, %0, %16\nv_max_u16 %1, %1, %16\nv_max_u16 %2, %2, %16\nv_max_u16 %3, %3, %16\nv_max_u16 %4, %4, %16\nv_max_u16 %5, %5, %16\nv_max_u16 %6, %6, %16\nv_max_u16 %7, %7, %16\nv_max_u16 %8, %8, %16\nv_max_u16 %9, %9, %16\nv_max_u16 %10, %10, %16\nv_max_u16 %11, %11, %16\nv_max_u16 %12, %12, %16\nv_max_u16 %13, %13, %16\nv_max_u16 %14, %14, %16\nv_max_u16 %15, %15, %16\n" : "+v"(a[0]),"+v"(a[1]),"+v"(a[2]),"+v"(a[3]),"+v"(a[4]),"+v"(a[5]),"+v"(a[6]),"+v"(a[7]),"+v"(a[8]),"+v"(a[9]),"+v"(a[10]),"+v"(a[11]),"+v"(a[12]),"+v"(a[13]),"+v"(a[14]),"+v"(a[15]) : "v"(b));
  unsigned s=0; for (int i=0;i<16;++i) s^=a[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }
__global__ void k_v_max_i16(unsigned* out, unsigned seed) {
  unsigned a[16]; for (int i=0;i<16;++i) a[i]=threadIdx.x*17+i+seed; unsigned b=seed|1;
  for (int it=0; it<ITER; ++it) asm volatile("v_max_i16 %0, %0, %16\nv_max_i16 %1, %1, %16\nv_max_i16 %2, %2, %16\nv_max_i16 %3, %3, %16\nv_max_i16 %4, %4, %16\nv_max_i16 %5, %5, %16\nv_max_i16 %6, %6, %16\nv_max_i16 %7, %7, %16\nv_max_i16 %8, %8, %16\nv_max_i16 %9, %9, %16\nv_max_i16 %10, %10, %16\nv_max_i16 %11, %11, %16\nv_max_i16 %12, %12, %16\nv_max_i16 %13, %13, %16\nv_max_i16 %14, %14, %16\nv_max_i16 %15, %15, %16\n" : "+v"(a[0]),"+v"(a[1]),"+v"(a[2]),"+v"(a[3]),"+v"(a[4]),"+v"(a[5]),"+v"(a[6]),"+v"(a[7]),"+v"(a[8]),"+v"(a[9]),"+v"(a[10]),"+v"(a[11]),"+v"(a[12]),"+v"(a[13]),"+v"(a[14]),"+v"(a[15]) : "v"(b));
  unsigned s=0; for (int i=0;i<16;++i) s^=a[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }
__global__ void k_v_add_u16(unsigned* out, unsigned seed) {
  unsigned a[16]; for (int i=0;i<16;++i) a[i]=threadIdx.x*17+i+seed; unsigned b=seed|1;
  for (int it=0; it<ITER; ++it) asm volatile("v_add_u16 %0, %0, %16\nv_add_u16 %1, %1, %16\nv_add_u16 %2, %2, %16\nv_add_u16 %3, %3, %16\nv_add_u16 %4, %4, %16\nv_add_u16 %5, %5, %16\nv_add_u16 %6, %6, %16\nv_add_u16 %7, %7, %16\nv_add_u16 %8, %8, %16\nv_add_u16 %9, %9, %16\nv_add_u16 %10, %10, %16\nv_add_u16 %11, %11, %16\nv_add_u16 %12, %12, %16\nv_add_u16 %13, %13, %16\nv_add_u16 %14, %14, %16\nv_add_u16 %15, %15, %16\n" : "+v"(a[0]),"+v"(a[1]),"+v"(a[2]),"+v"(a[3]),"+v"(a[4]),"+v"(a[5]),"+v"(a[6]),"+v"(a[7]),"+v"(a[8]),"+v"(a[9]),"+v"(a[10]),"+v"(a[11]),"+v"(a[12]),"+v"(a[13]),"+v"(a[14]),"+v"(a[15]) : "v"(b));
  unsigned s=0; for (int i=0;i<16;++i) s^=a[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }
__global__ void k_v_lshlrev_b32(unsigned* out, unsigned seed) {
  unsigned a[16]; for (int i=0;i<16;++i) a[i]=threadIdx.x*17+i+seed; unsigned b=seed|1;
  for (int it=0; it<ITER; ++it) asm volatile("v_lshlrev_b32 %0, %0, %16\nv_lshlrev_b32 %1, %1, %16\nv_lshlrev_b32 %2, %2, %16\nv_lshlrev_b32 %3, %3, %16\nv_lshlrev_b32 %4, %4, %16\nv_lshlrev_b32 %5, %5, %16\nv_lshlrev_b32 %6, %6, %16\nv_lshlrev_b32 %7, %7, %16\nv_lshlrev_b32 %8, %8, %16\nv_lshlrev_b32 %9, %9, %16\nv_lshlrev_b32 %10, %10, %16\nv_lshlrev_b32 %11, %11, %16\nv_lshlrev_b32 %12, %12, %16\nv_lshlrev_b32 %13, %13, %16\nv_lshlrev_b32 %14, %14, %16\nv_lshlrev_b32 %15, %15, %16\n" : "+v"(a[0]),"+v"(a[1]),"+v"(a[2]),"+v"(a[3]),"+v"(a[4]),"+v"(a[5]),"+v"(a[6]),"+v"(a[7]),"+v"(a[8]),"+v"(a[9]),"+v"(a[10]),"+v"(a[11]),"+v"(a[12]),"+v"(a[13]),"+v"(a[14]),"+v"(a[15]) : "v"(b));
  unsigned s=0; for (int i=0;i<16;++i) s^=a[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }
__global__ void k_v_pk_add_f32x(unsigned* out, unsigned seed) {
  double a[8]; for (int i=0;i<8;++i) a[i]=threadIdx.x*17+i+seed; double b=seed|1;
  for (int it=0; it<ITER; ++it) asm volatile("v_pk_add_f32 %0, %0, %8\nv_pk_add_f32 %1, %1, %8\nv_pk_add_f32 %2, %2, %8\nv_pk_add_f32 %3, %3, %8\nv_pk_add_f32 %4, %4, %8\nv_pk_add_f32 %5, %5, %8\nv_pk_add_f32 %6, %6, %8\nv_pk_add_f32 %7, %7, %8\nv_pk_add_f32 %0, %0, %8\nv_pk_add_f32 %1, %1, %8\nv_pk_add_f32 %2, %2, %8\nv_pk_add_f32 %3, %3, %8\nv_pk_add_f32 %4, %4, %8\nv_pk_add_f32 %5, %5, %8\nv_pk_add_f32 %6, %6, %8\nv_pk_add_f32 %7, %7, %8\n" : "+v"(a[0]),"+v"(a[1]),"+v"(a[2]),"+v"(a[3]),"+v"(a[4]),"+v"(a[5]),"+v"(a[6]),"+v"(a[7]) : "v"(b));
  double s=0; for (int i=0;i<8;++i) s+=a[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=(unsigned)s; }
__global__ void k_v_max3_i32(unsigned* out, unsigned seed) {
  unsigned a[16]; for (int i=0;i<16;++i) a[i]=threadIdx.x*17+i+seed; unsigned b=seed|1, c=seed+3;
  for (int it=0; it<ITER; ++it) asm volatile("v_max3_i32 %0, %0, %16, %17\nv_max3_i32 %1, %1, %16, %17\nv_max3_i32 %2, %2, %16, %17\nv_max3_i32 %3, %3, %16, %17\nv_max3_i32 %4, %4, %16, %17\nv_max3_i32 %5, %5, %16, %17\nv_max3_i32 %6, %6, %16, %17\nv_max3_i32 %7, %7, %16, %17\nv_max3_i32 %8, %8, %16, %17\nv_max3_i32 %9, %9, %16, %17\nv_max3_i32 %10, %10, %16, %17\nv_max3_i32 %11, %11, %16, %17\nv_max3_i32 %12, %12, %16, %17\nv_max3_i32 %13, %13, %16, %17\nv_max3_i32 %14, %14, %16, %17\nv_max3_i32 %15, %15, %16, %17\n" : "+v"(a[0]),"+v"(a[1]),"+v"(a[2]),"+v"(a[3]),"+v"(a[4]),"+v"(a[5]),"+v"(a[6]),"+v"(a[7]),"+v"(a[8]),"+v"(a[9]),"+v"(a[10]),"+v"(a[11]),"+v"(a[12]),"+v"(a[13]),"+v"(a[14]),"+v"(a[15]) : "v"(b),"v"(c));
  unsigned s=0; for (int i=0;i<16;++i) s^=a[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }
__global__ void k_v_max3_u32(unsigned* out, unsigned seed) {
  unsigned a[16]; for (int i=0;i<16;++i) a[i]=threadIdx.x*17+i+seed; unsigned b=seed|1, c=seed+3;
  for (int it=0; it<ITER; ++it) asm volatile("v_max3_u32 %0, %0, %16, %17\nv_max3_u32 %1, %1, %16, %17\nv_max3_u32 %2, %2, %16, %17\nv_max3_u32 %3, %3, %16, %17\nv_max3_u32 %4, %4, %16, %17\nv_max3_u32 %5, %5, %16, %17\nv_max3_u32 %6, %6, %16, %17\nv_max3_u32 %7, %7, %16, %17\nv_max3_u32 %8, %8, %16, %17\nv_max3_u32 %9, %9, %16, %17\nv_max3_u32 %10, %10, %16, %17\nv_max3_u32 %11, %11, %16, %17\nv_max3_u32 %12, %12, %16, %17\nv_max3_u32 %13, %13, %16, %17\nv_max3_u32 %14, %14, %16, %17\nv_max3_u32 %15, %15, %16, %17\n" : "+v"(a[0]),"+v"(a[1]),"+v"(a[2]),"+v"(a[3]),"+v"(a[4]),"+v"(a[5]),"+v"(a[6]),"+v"(a[7]),"+v"(a[8]),"+v"(a[9]),"+v"(a[10]),"+v"(a[11]),"+v"(a[12]),"+v"(a[13]),"+v"(a[14]),"+v"(a[15]) : "v"(b),"v"(c));
  unsigned s=0; for (int i=0;i<16;++i) s^=a[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }
__global__ void k_v_max3_f32(unsigned* out, unsigned seed) {
  unsigned a[16]; for (int i=0;i<16;++i) a[i]=threadIdx.x*17+i+seed; unsigned b=seed|1, c=seed+3;
  for (int it=0; it<ITER; ++it) asm volatile("v_max3_f32 %0, %0, %16, %17\nv_max3_f32 %1, %1, %16, %17\nv_max3_f32 %2, %2, %16, %17\nv_max3_f32 %3, %3, %16, %17\nv_max3_f32 %4, %4, %16, %17\nv_max3_f32 %5, %5, %16, %17\nv_max3_f32 %6, %6, %16, %17\nv_max3_f32 %7, %7, %16, %17\nv_max3_f32 %8, %8, %16, %17\nv_max3_f32 %9, %9, %16, %17\nv_max3_f32 %10, %10, %16, %17\nv_max3_f32 %11, %11, %16, %17\nv_max3_f32 %12, %12, %16, %17\nv_max3_f32 %13, %13, %16, %17\nv_max3_f32 %14, %14, %16, %17\nv_max3_f32 %15, %15, %16, %17\n" : "+v"(a[0]),"+v"(a[1]),"+v"(a[2]),"+v"(a[3]),"+v"(a[4]),"+v"(a[5]),"+v"(a[6]),"+v"(a[7]),"+v"(a[8]),"+v"(a[9]),"+v"(a[10]),"+v"(a[11]),"+v"(a[12]),"+v"(a[13]),"+v"(a[14]),"+v"(a[15]) : "v"(b),"v"(c));
  unsigned s=0; for (int i=0;i<16;++i) s^=a[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }
__global__ void k_v_maximum3_f32(unsigned* out, unsigned seed) {
  unsigned a[16]; for (int i=0;i<16;++i) a[i]=threadIdx.x*17+i+seed; unsigned b=seed|1, c=seed+3;
  for (int it=0; it<ITER; ++it) asm volatile("v_maximum3_f32 %0, %0, %16, %17\nv_maximum3_f32 %1, %1, %16, %17\nv_maximum3_f32 %2, %2, %16, %17\nv_maximum3_f32 %3, %3, %16, %17\nv_maximum3_f32 %4, %4, %16, %17\nv_maximum3_f32 %5, %5, %16, %17\nv_maximum3_f32 %6, %6, %16, %17\nv_maximum3_f32 %7, %7, %16, %17\nv_maximum3_f32 %8, %8, %16, %17\nv_maximum3_f32 %9, %9, %16, %17\nv_maximum3_f32 %10, %10, %16, %17\nv_maximum3_f32 %11, %11, %16, %17\nv_maximum3_f32 %12, %12, %16, %17\nv_maximum3_f32 %13, %13, %16, %17\nv_maximum3_f32 %14, %14, %16, %17\nv_maximum3_f32 %15, %15, %16, %17\n" : "+v"(a[0]),"+v"(a[1]),"+v"(a[2]),"+v"(a[3]),"+v"(a[4]),"+v"(a[5]),"+v"(a[6]),"+v"(a[7]),"+v"(a[8]),"+v"(a[9]),"+v"(a[10]),"+v"(a[11]),"+v"(a[12]),"+v"(a[13]),"+v"(a[14]),"+v"(a[15]) : "v"(b),"v"(c));
  unsigned s=0; for (int i=0;i<16;++i) s^=a[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }
__global__ void k_v_pk_maximum3_f16(unsigned* out, unsigned seed) {
  unsigned a[16]; for (int i=0;i<16;++i) a[i]=threadIdx.x*17+i+seed; unsigned b=seed|1, c=seed+3;
  for (int it=0; it<ITER; ++it) asm volatile("v_pk_maximum3_f16 %0, %0, %16, %17\nv_pk_maximum3_f16 %1, %1, %16, %17\nv_pk_maximum3_f16 %2, %2, %16, %17\nv_pk_maximum3_f16 %3, %3, %16, %17\nv_pk_maximum3_f16 %4, %4, %16, %17\nv_pk_maximum3_f16 %5, %5, %16, %17\nv_pk_maximum3_f16 %6, %6, %16, %17\nv_pk_maximum3_f16 %7, %7, %16, %17\nv_pk_maximum3_f16 %8, %8, %16, %17\nv_pk_maximum3_f16 %9, %9, %16, %17\nv_pk_maximum3_f16 %10, %10, %16, %17\nv_pk_maximum3_f16 %11, %11, %16, %17\nv_pk_maximum3_f16 %12, %12, %16, %17\nv_pk_maximum3_f16 %13, %13, %16, %17\nv_pk_maximum3_f16 %14, %14, %16, %17\nv_pk_maximum3_f16 %15, %15, %16, %17\n" : "+v"(a[0]),"+v"(a[1]),"+v"(a[2]),"+v"(a[3]),"+v"(a[4]),"+v"(a[5]),"+v"(a[6]),"+v"(a[7]),"+v"(a[8]),"+v"(a[9]),"+v"(a[10]),"+v"(a[11]),"+v"(a[12]),"+v"(a[13]),"+v"(a[14]),"+v"(a[15]) : "v"(b),"v"(c));
  unsigned s=0; for (int i=0;i<16;++i) s^=a[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }
__global__ void k_v_add3_u32(unsigned* out, unsigned seed) {
  unsigned a[16]; for (int i=0;i<16;++i) a[i]=threadIdx.x*17+i+seed; unsigned b=seed|1, c=seed+3;
  for (int it=0; it<ITER; ++it) asm volatile("v_add3_u32 %0, %0, %16, %17\nv_add3_u32 %1, %1, %16, %17\nv_add3_u32 %2, %2, %16, %17\nv_add3_u32 %3, %3, %16, %17\nv_add3_u32 %4, %4, %16, %17\nv_add3_u32 %5, %5, %16, %17\nv_add3_u32 %6, %6, %16, %17\nv_add3_u32 %7, %7, %16, %17\nv_add3_u32 %8, %8, %16, %17\nv_add3_u32 %9, %9, %16, %17\nv_add3_u32 %10, %10, %16, %17\nv_add3_u32 %11, %11, %16, %17\nv_add3_u32 %12, %12, %16, %17\nv_add3_u32 %13, %13, %16, %17\nv_add3_u32 %14, %14, %16, %17\nv_add3_u32 %15, %15, %16, %17\n" : "+v"(a[0]),"+v"(a[1]),"+v"(a[2]),"+v"(a[3]),"+v"(a[4]),"+v"(a[5]),"+v"(a[6]),"+v"(a[7]),"+v"(a[8]),"+v"(a[9]),"+v"(a[10]),"+v"(a[11]),"+v"(a[12]),"+v"(a[13]),"+v"(a[14]),"+v"(a[15]) : "v"(b),"v"(c));
  unsigned s=0; for (int i=0;i<16;++i) s^=a[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }
__global__ void k_v_lshl_add_u32(unsigned* out, unsigned seed) {
  unsigned a[16]; for (int i=0;i<16;++i) a[i]=threadIdx.x*17+i+seed; unsigned b=seed|1, c=seed+3;
  for (int it=0; it<ITER; ++it) asm volatile("v_lshl_add_u32 %0, %0, %16, %17\nv_lshl_add_u32 %1, %1, %16, %17\nv_lshl_add_u32 %2, %2, %16, %17\nv_lshl_add_u32 %3, %3, %16, %17\nv_lshl_add_u32 %4, %4, %16, %17\nv_lshl_add_u32 %5, %5, %16, %17\nv_lshl_add_u32 %6, %6, %16, %17\nv_lshl_add_u32 %7, %7, %16, %17\nv_lshl_add_u32 %8, %8, %16, %17\nv_lshl_add_u32 %9, %9, %16, %17\nv_lshl_add_u32 %10, %10, %16, %17\nv_lshl_add_u32 %11, %11, %16, %17\nv_lshl_add_u32 %12, %12, %16, %17\nv_lshl_add_u32 %13, %13, %16, %17\nv_lshl_add_u32 %14, %14, %16, %17\nv_lshl_add_u32 %15, %15, %16, %17\n" : "+v"(a[0]),"+v"(a[1]),"+v"(a[2]),"+v"(a[3]),"+v"(a[4]),"+v"(a[5]),"+v"(a[6]),"+v"(a[7]),"+v"(a[8]),"+v"(a[9]),"+v"(a[10]),"+v"(a[11]),"+v"(a[12]),"+v"(a[13]),"+v"(a[14]),"+v"(a[15]) : "v"(b),"v"(c));
  unsigned s=0; for (int i=0;i<16;++i) s^=a[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }
__global__ void k_v_med3_i32(unsigned* out, unsigned seed) {
  unsigned a[16]; for (int i=0;i<16;++i) a[i]=threadIdx.x*17+i+seed; unsigned b=seed|1, c=seed+3;
  for (int it=0; it<ITER; ++it) asm volatile("v_med3_i32 %0, %0, %16, %17\nv_med3_i32 %1, %1, %16, %17\nv_med3_i32 %2, %2, %16, %17\nv_med3_i32 %3, %3, %16, %17\nv_med3_i32 %4, %4, %16, %17\nv_med3_i32 %5, %5, %16, %17\nv_med3_i32 %6, %6, %16, %17\nv_med3_i32 %7, %7, %16, %17\nv_med3_i32 %8, %8, %16, %17\nv_med3_i32 %9, %9, %16, %17\nv_med3_i32 %10, %10, %16, %17\nv_med3_i32 %11, %11, %16, %17\nv_med3_i32 %12, %12, %16, %17\nv_med3_i32 %13, %13, %16, %17\nv_med3_i32 %14, %14, %16, %17\nv_med3_i32 %15, %15, %16, %17\n" : "+v"(a[0]),"+v"(a[1]),"+v"(a[2]),"+v"(a[3]),"+v"(a[4]),"+v"(a[5]),"+v"(a[6]),"+v"(a[7]),"+v"(a[8]),"+v"(a[9]),"+v"(a[10]),"+v"(a[11]),"+v"(a[12]),"+v"(a[13]),"+v"(a[14]),"+v"(a[15]) : "v"(b),"v"(c));
  unsigned s=0; for (int i=0;i<16;++i) s^=a[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }
__global__ void k_v_perm_b32(unsigned* out, unsigned seed) {
  unsigned a[16]; for (int i=0;i<16;++i) a[i]=threadIdx.x*17+i+seed; unsigned b=seed|1, c=seed+3;
  for (int it=0; it<ITER; ++it) asm volatile("v_perm_b32 %0, %0, %16, %17\nv_perm_b32 %1, %1, %16, %17\nv_perm_b32 %2, %2, %16, %17\nv_perm_b32 %3, %3, %16, %17\nv_perm_b32 %4, %4, %16, %17\nv_perm_b32 %5, %5, %16, %17\nv_perm_b32 %6, %6, %16, %17\nv_perm_b32 %7, %7, %16, %17\nv_perm_b32 %8, %8, %16, %17\nv_perm_b32 %9, %9, %16, %17\nv_perm_b32 %10, %10, %16, %17\nv_perm_b32 %11, %11, %16, %17\nv_perm_b32 %12, %12, %16, %17\nv_perm_b32 %13, %13, %16, %17\nv_perm_b32 %14, %14, %16, %17\nv_perm_b32 %15, %15, %16, %17\n" : "+v"(a[0]),"+v"(a[1]),"+v"(a[2]),"+v"(a[3]),"+v"(a[4]),"+v"(a[5]),"+v"(a[6]),"+v"(a[7]),"+v"(a[8]),"+v"(a[9]),"+v"(a[10]),"+v"(a[11]),"+v"(a[12]),"+v"(a[13]),"+v"(a[14]),"+v"(a[15]) : "v"(b),"v"(c));
  unsigned s=0; for (int i=0;i<16;++i) s^=a[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }
__global__ void k_v_pk_mad_u16(unsigned* out, unsigned seed) {
  unsigned a[16]; for (int i=0;i<16;++i) a[i]=threadIdx.x*17+i+seed; unsigned b=seed|1, c=seed+3;
  for (int it=0; it<ITER; ++it) asm volatile("v_pk_mad_u16 %0, %0, %16, %17\nv_pk_mad_u16 %1, %1, %16, %17\nv_pk_mad_u16 %2, %2, %16, %17\nv_pk_mad_u16 %3, %3, %16, %17\nv_pk_mad_u16 %4, %4, %16, %17\nv_pk_mad_u16 %5, %5, %16, %17\nv_pk_mad_u16 %6, %6, %16, %17\nv_pk_mad_u16 %7, %7, %16, %17\nv_pk_mad_u16 %8, %8, %16, %17\nv_pk_mad_u16 %9, %9, %16, %17\nv_pk_mad_u16 %10, %10, %16, %17\nv_pk_mad_u16 %11, %11, %16, %17\nv_pk_mad_u16 %12, %12, %16, %17\nv_pk_mad_u16 %13, %13, %16, %17\nv_pk_mad_u16 %14, %14, %16, %17\nv_pk_mad_u16 %15, %15, %16, %17\n" : "+v"(a[0]),"+v"(a[1]),"+v"(a[2]),"+v"(a[3]),"+v"(a[4]),"+v"(a[5]),"+v"(a[6]),"+v"(a[7]),"+v"(a[8]),"+v"(a[9]),"+v"(a[10]),"+v"(a[11]),"+v"(a[12]),"+v"(a[13]),"+v"(a[14]),"+v"(a[15]) : "v"(b),"v"(c));
  unsigned s=0; for (int i=0;i<16;++i) s^=a[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }
__global__ void k_v_mad_u32_u24(unsigned* out, unsigned seed) {
  unsigned a[16]; for (int i=0;i<16;++i) a[i]=threadIdx.x*17+i+seed; unsigned b=seed|1, c=seed+3;
  for (int it=0; it<ITER; ++it) asm volatile("v_mad_u32_u24 %0, %0, %16, %17\nv_mad_u32_u24 %1, %1, %16, %17\nv_mad_u32_u24 %2, %2, %16, %17\nv_mad_u32_u24 %3, %3, %16, %17\nv_mad_u32_u24 %4, %4, %16, %17\nv_mad_u32_u24 %5, %5, %16, %17\nv_mad_u32_u24 %6, %6, %16, %17\nv_mad_u32_u24 %7, %7, %16, %17\nv_mad_u32_u24 %8, %8, %16, %17\nv_mad_u32_u24 %9, %9, %16, %17\nv_mad_u32_u24 %10, %10, %16, %17\nv_mad_u32_u24 %11, %11, %16, %17\nv_mad_u32_u24 %12, %12, %16, %17\nv_mad_u32_u24 %13, %13, %16, %17\nv_mad_u32_u24 %14, %14, %16, %17\nv_mad_u32_u24 %15, %15, %16, %17\n" : "+v"(a[0]),"+v"(a[1]),"+v"(a[2]),"+v"(a[3]),"+v"(a[4]),"+v"(a[5]),"+v"(a[6]),"+v"(a[7]),"+v"(a[8]),"+v"(a[9]),"+v"(a[10]),"+v"(a[11]),"+v"(a[12]),"+v"(a[13]),"+v"(a[14]),"+v"(a[15]) : "v"(b),"v"(c));
  unsigned s=0; for (int i=0;i<16;++i) s^=a[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }
__global__ void k_v_and_or_b32(unsigned* out, unsigned seed) {
  unsigned a[16]; for (int i=0;i<16;++i) a[i]=threadIdx.x*17+i+seed; unsigned b=seed|1, c=seed+3;
  for (int it=0; it<ITER; ++it) asm volatile("v_and_or_b32 %0, %0, %16, %17\nv_and_or_b32 %1, %1, %16, %17\nv_and_or_b32 %2, %2, %16, %17\nv_and_or_b32 %3, %3, %16, %17\nv_and_or_b32 %4, %4, %16, %17\nv_and_or_b32 %5, %5, %16, %17\nv_and_or_b32 %6, %6, %16, %17\nv_and_or_b32 %7, %7, %16, %17\nv_and_or_b32 %8, %8, %16, %17\nv_and_or_b32 %9, %9, %16, %17\nv_and_or_b32 %10, %10, %16, %17\nv_and_or_b32 %11, %11, %16, %17\nv_and_or_b32 %12, %12, %16, %17\nv_and_or_b32 %13, %13, %16, %17\nv_and_or_b32 %14, %14, %16, %17\nv_and_or_b32 %15, %15, %16, %17\n" : "+v"(a[0]),"+v"(a[1]),"+v"(a[2]),"+v"(a[3]),"+v"(a[4]),"+v"(a[5]),"+v"(a[6]),"+v"(a[7]),"+v"(a[8]),"+v"(a[9]),"+v"(a[10]),"+v"(a[11]),"+v"(a[12]),"+v"(a[13]),"+v"(a[14]),"+v"(a[15]) : "v"(b),"v"(c));
  unsigned s=0; for (int i=0;i<16;++i) s^=a[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }
__global__ void k_v_max3_i16(unsigned* out, unsigned seed) {
  unsigned a[16]; for (int i=0;i<16;++i) a[i]=threadIdx.x*17+i+seed; unsigned b=seed|1, c=seed+3;
  for (int it=0; it<ITER; ++it) asm volatile("v_max3_i16 %0, %0, %16, %17\nv_max3_i16 %1, %1, %16, %17\nv_max3_i16 %2, %2, %16, %17\nv_max3_i16 %3, %3, %16, %17\nv_max3_i16 %4, %4, %16, %17\nv_max3_i16 %5, %5, %16, %17\nv_max3_i16 %6, %6, %16, %17\nv_max3_i16 %7, %7, %16, %17\nv_max3_i16 %8, %8, %16, %17\nv_max3_i16 %9, %9, %16, %17\nv_max3_i16 %10, %10, %16, %17\nv_max3_i16 %11, %11, %16, %17\nv_max3_i16 %12, %12, %16, %17\nv_max3_i16 %13, %13, %16, %17\nv_max3_i16 %14, %14, %16, %17\nv_max3_i16 %15, %15, %16, %17\n" : "+v"(a[0]),"+v"(a[1]),"+v"(a[2]),"+v"(a[3]),"+v"(a[4]),"+v"(a[5]),"+v"(a[6]),"+v"(a[7]),"+v"(a[8]),"+v"(a[9]),"+v"(a[10]),"+v"(a[11]),"+v"(a[12]),"+v"(a[13]),"+v"(a[14]),"+v"(a[15]) : "v"(b),"v"(c));
  unsigned s=0; for (int i=0;i<16;++i) s^=a[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }
__global__ void k_v_max3_u16(unsigned* out, unsigned seed) {
  unsigned a[16]; for (int i=0;i<16;++i) a[i]=threadIdx.x*17+i+seed; unsigned b=seed|1, c=seed+3;
  for (int it=0; it<ITER; ++it) asm volatile("v_max3_u16 %0, %0, %16, %17\nv_max3_u16 %1, %1, %16, %17\nv_max3_u16 %2, %2, %16, %17\nv_max3_u16 %3, %3, %16, %17\nv_max3_u16 %4, %4, %16, %17\nv_max3_u16 %5, %5, %16, %17\nv_max3_u16 %6, %6, %16, %17\nv_max3_u16 %7, %7, %16, %17\nv_max3_u16 %8, %8, %16, %17\nv_max3_u16 %9, %9, %16, %17\nv_max3_u16 %10, %10, %16, %17\nv_max3_u16 %11, %11, %16, %17\nv_max3_u16 %12, %12, %16, %17\nv_max3_u16 %13, %13, %16, %17\nv_max3_u16 %14, %14, %16, %17\nv_max3_u16 %15, %15, %16, %17\n" : "+v"(a[0]),"+v"(a[1]),"+v"(a[2]),"+v"(a[3]),"+v"(a[4]),"+v"(a[5]),"+v"(a[6]),"+v"(a[7]),"+v"(a[8]),"+v"(a[9]),"+v"(a[10]),"+v"(a[11]),"+v"(a[12]),"+v"(a[13]),"+v"(a[14]),"+v"(a[15]) : "v"(b),"v"(c));
  unsigned s=0; for (int i=0;i<16;++i) s^=a[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }
__global__ void k_v_max3_f16(unsigned* out, unsigned seed) {
  unsigned a[16]; for (int i=0;i<16;++i) a[i]=threadIdx.x*17+i+seed; unsigned b=seed|1, c=seed+3;
  for (int it=0; it<ITER; ++it) asm volatile("v_max3_f16 %0, %0, %16, %17\nv_max3_f16 %1, %1, %16, %17\nv_max3_f16 %2, %2, %16, %17\nv_max3_f16 %3, %3, %16, %17\nv_max3_f16 %4, %4, %16, %17\nv_max3_f16 %5, %5, %16, %17\nv_max3_f16 %6, %6, %16, %17\nv_max3_f16 %7, %7, %16, %17\nv_max3_f16 %8, %8, %16, %17\nv_max3_f16 %9, %9, %16, %17\nv_max3_f16 %10, %10, %16, %17\nv_max3_f16 %11, %11, %16, %17\nv_max3_f16 %12, %12, %16, %17\nv_max3_f16 %13, %13, %16, %17\nv_max3_f16 %14, %14, %16, %17\nv_max3_f16 %15, %15, %16, %17\n" : "+v"(a[0]),"+v"(a[1]),"+v"(a[2]),"+v"(a[3]),"+v"(a[4]),"+v"(a[5]),"+v"(a[6]),"+v"(a[7]),"+v"(a[8]),"+v"(a[9]),"+v"(a[10]),"+v"(a[11]),"+v"(a[12]),"+v"(a[13]),"+v"(a[14]),"+v"(a[15]) : "v"(b),"v"(c));
  unsigned s=0; for (int i=0;i<16;++i) s^=a[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }
__global__ void k_v_bfe_u32(unsigned* out, unsigned seed) {
  unsigned a[16]; for (int i=0;i<16;++i) a[i]=threadIdx.x*17+i+seed; unsigned b=seed|1, c=seed+3;
  for (int it=0; it<ITER; ++it) asm volatile("v_bfe_u32 %0, %0, %16, %17\nv_bfe_u32 %1, %1, %16, %17\nv_bfe_u32 %2, %2, %16, %17\nv_bfe_u32 %3, %3, %16, %17\nv_bfe_u32 %4, %4, %16, %17\nv_bfe_u32 %5, %5, %16, %17\nv_bfe_u32 %6, %6, %16, %17\nv_bfe_u32 %7, %7, %16, %17\nv_bfe_u32 %8, %8, %16, %17\nv_bfe_u32 %9, %9, %16, %17\nv_bfe_u32 %10, %10, %16, %17\nv_bfe_u32 %11, %11, %16, %17\nv_bfe_u32 %12, %12, %16, %17\nv_bfe_u32 %13, %13, %16, %17\nv_bfe_u32 %14, %14, %16, %17\nv_bfe_u32 %15, %15, %16, %17\n" : "+v"(a[0]),"+v"(a[1]),"+v"(a[2]),"+v"(a[3]),"+v"(a[4]),"+v"(a[5]),"+v"(a[6]),"+v"(a[7]),"+v"(a[8]),"+v"(a[9]),"+v"(a[10]),"+v"(a[11]),"+v"(a[12]),"+v"(a[13]),"+v"(a[14]),"+v"(a[15]) : "v"(b),"v"(c));
  unsigned s=0; for (int i=0;i<16;++i) s^=a[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }
__global__ void k_v_alignbit_b32(unsigned* out, unsigned seed) {
  unsigned a[16]; for (int i=0;i<16;++i) a[i]=threadIdx.x*17+i+seed; unsigned b=seed|1, c=seed+3;
  for (int it=0; it<ITER; ++it) asm volatile("v_alignbit_b32 %0, %0, %16, %17\nv_alignbit_b32 %1, %1, %16, %17\nv_alignbit_b32 %2, %2, %16, %17\nv_alignbit_b32 %3, %3, %16, %17\nv_alignbit_b32 %4, %4, %16, %17\nv_alignbit_b32 %5, %5, %16, %17\nv_alignbit_b32 %6, %6, %16, %17\nv_alignbit_b32 %7, %7, %16, %17\nv_alignbit_b32 %8, %8, %16, %17\nv_alignbit_b32 %9, %9, %16, %17\nv_alignbit_b32 %10, %10, %16, %17\nv_alignbit_b32 %11, %11, %16, %17\nv_alignbit_b32 %12, %12, %16, %17\nv_alignbit_b32 %13, %13, %16, %17\nv_alignbit_b32 %14, %14, %16, %17\nv_alignbit_b32 %15, %15, %16, %17\n" : "+v"(a[0]),"+v"(a[1]),"+v"(a[2]),"+v"(a[3]),"+v"(a[4]),"+v"(a[5]),"+v"(a[6]),"+v"(a[7]),"+v"(a[8]),"+v"(a[9]),"+v"(a[10]),"+v"(a[11]),"+v"(a[12]),"+v"(a[13]),"+v"(a[14]),"+v"(a[15]) : "v"(b),"v"(c));
  unsigned s=0; for (int i=0;i<16;++i) s^=a[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }
__global__ void k_v_xad_u32(unsigned* out, unsigned seed) {
  unsigned a[16]; for (int i=0;i<16;++i) a[i]=threadIdx.x*17+i+seed; unsigned b=seed|1, c=seed+3;
  for (int it=0; it<ITER; ++it) asm volatile("v_xad_u32 %0, %0, %16, %17\nv_xad_u32 %1, %1, %16, %17\nv_xad_u32 %2, %2, %16, %17\nv_xad_u32 %3, %3, %16, %17\nv_xad_u32 %4, %4, %16, %17\nv_xad_u32 %5, %5, %16, %17\nv_xad_u32 %6, %6, %16, %17\nv_xad_u32 %7, %7, %16, %17\nv_xad_u32 %8, %8, %16, %17\nv_xad_u32 %9, %9, %16, %17\nv_xad_u32 %10, %10, %16, %17\nv_xad_u32 %11, %11, %16, %17\nv_xad_u32 %12, %12, %16, %17\nv_xad_u32 %13, %13, %16, %17\nv_xad_u32 %14, %14, %16, %17\nv_xad_u32 %15, %15, %16, %17\n" : "+v"(a[0]),"+v"(a[1]),"+v"(a[2]),"+v"(a[3]),"+v"(a[4]),"+v"(a[5]),"+v"(a[6]),"+v"(a[7]),"+v"(a[8]),"+v"(a[9]),"+v"(a[10]),"+v"(a[11]),"+v"(a[12]),"+v"(a[13]),"+v"(a[14]),"+v"(a[15]) : "v"(b),"v"(c));
  unsigned s=0; for (int i=0;i<16;++i) s^=a[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }

template <typename K> void run(K kern, const char* name, int w, unsigned* d_out) {
  int blocks = 256*w; hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_out, 1u); hipDeviceSynchronize();
  hipEventRecord(e0); hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_out, 2u); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms=0; hipEventElapsedTime(&ms,e0,e1);
  double per = (double)blocks*4*ITER*16/1024.0/(ms*1e-3);
  printf("%-22s waves/SIMD=%d %.3f ms => %.2f cycles/instr @2.4GHz\n", name, w, ms, 2.4e9/per);
}
int main() { unsigned* d; hipMalloc(&d, 256*8*256*4);
  run(k_v_pk_add_u16, "v_pk_add_u16", 4, d);
  run(k_v_pk_max_i16, "v_pk_max_i16", 4, d);
  run(k_v_pk_max_u16, "v_pk_max_u16", 4, d);
  run(k_v_pk_min_u16, "v_pk_min_u16", 4, d);
  run(k_v_add_u32, "v_add_u32", 4, d);
  run(k_v_sub_u32, "v_sub_u32", 4, d);
  run(k_v_max_i32, "v_max_i32", 4, d);
  run(k_v_max_u32, "v_max_u32", 4, d);
  run(k_v_min_u32, "v_min_u32", 4, d);
  run(k_v_xor_b32, "v_xor_b32", 4, d);
  run(k_v_and_b32, "v_and_b32", 4, d);
  run(k_v_max_f32, "v_max_f32", 4, d);
  run(k_v_add_f32, "v_add_f32", 4, d);
  run(k_v_pk_add_f16, "v_pk_add_f16", 4, d);
  run(k_v_pk_max_f16, "v_pk_max_f16", 4, d);
  run(k_v_max_u16, "v_max_u16", 4, d);
  run(k_v_max_i16, "v_max_i16", 4, d);
  run(k_v_add_u16, "v_add_u16", 4, d);
  run(k_v_lshlrev_b32, "v_lshlrev_b32", 4, d);
  run(k_v_pk_add_f32x, "v_pk_add_f32x", 4, d);
  run(k_v_max3_i32, "v_max3_i32", 4, d);
  run(k_v_max3_u32, "v_max3_u32", 4, d);
  run(k_v_max3_f32, "v_max3_f32", 4, d);
  run(k_v_maximum3_f32, "v_maximum3_f32", 4, d);
  run(k_v_pk_maximum3_f16, "v_pk_maximum3_f16", 4, d);
  run(k_v_add3_u32, "v_add3_u32", 4, d);
  run(k_v_lshl_add_u32, "v_lshl_add_u32", 4, d);
  run(k_v_med3_i32, "v_med3_i32", 4, d);
  run(k_v_perm_b32, "v_perm_b32", 4, d);
  run(k_v_pk_mad_u16, "v_pk_mad_u16", 4, d);
  run(k_v_mad_u32_u24, "v_mad_u32_u24", 4, d);
  run(k_v_and_or_b32, "v_and_or_b32", 4, d);
  run(k_v_max3_i16, "v_max3_i16", 4, d);
  run(k_v_max3_u16, "v_max3_u16", 4, d);
  run(k_v_max3_f16, "v_max3_f16", 4, d);
  run(k_v_bfe_u32, "v_bfe_u32", 4, d);
  run(k_v_alignbit_b32, "v_alignbit_b32", 4, d);
  run(k_v_xad_u32, "v_xad_u32", 4, d);
  return 0; }
