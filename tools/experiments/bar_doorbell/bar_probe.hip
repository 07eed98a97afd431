// Can the host write a doorbell that lives in DEVICE memory (through the PCIe BAR), so that a resident kernel polls local memory instead of host memory?
// Probe 1: hipExtMallocWithFlags(hipDeviceMallocFinegrained) and a plain host store.  Probe 2: the HSA route (fine-grained pool of the GPU agent, access allowed to the CPU agent).
// Prints the round trip of: host writes sequence number -> kernel sees it -> kernel writes it to a host-mapped word -> host sees it.
//   hipcc --offload-arch=gfx950 -O2 bar_probe.hip -o bar_probe -lhsa-runtime64 && ./bar_probe
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <csignal>
#include <csetjmp>
#include <atomic>

__global__ void echo(volatile unsigned* door, volatile unsigned* answer, unsigned last)
{
	unsigned seen = 0;
	while (seen != last) {
		const unsigned v = __atomic_load_n((unsigned*)door, __ATOMIC_RELAXED);
		if (v != seen) { seen = v; __atomic_store_n((unsigned*)answer, v, __ATOMIC_RELAXED); __threadfence_system(); }
	}
}

static sigjmp_buf jb;
static void on_segv(int) { siglongjmp(jb, 1); }

static double run(volatile unsigned* door, const char* what, bool door_is_device)
{
	unsigned* answer = nullptr;
	hipHostMalloc((void**)&answer, 64, hipHostMallocMapped);
	*answer = 0;
	unsigned* d_answer = nullptr; hipHostGetDevicePointer((void**)&d_answer, answer, 0);
	const unsigned N = 20000;
	hipStream_t s; hipStreamCreate(&s);
	hipLaunchKernelGGL(echo, dim3(1), dim3(1), 0, s, door, (volatile unsigned*)d_answer, N);
	const auto t0 = std::chrono::steady_clock::now();
	bool gave_up = false;
	for (unsigned i = 1; i <= N && !gave_up; ++i) {
		__atomic_store_n((unsigned*)door, i, __ATOMIC_RELEASE);
		if (door_is_device) __builtin_ia32_sfence();      // (the BAR is mapped write-combining: the store sits in the WC buffer until it is flushed)
		const auto w0 = std::chrono::steady_clock::now();
		while (__atomic_load_n(answer, __ATOMIC_ACQUIRE) != i) {
			if (std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count() > 2.0) { gave_up = true; printf("  the kernel never saw value %u: giving up\n", i); break; }
		}
	}
	if (gave_up) { hipStream_t s2; hipStreamCreateWithFlags(&s2, hipStreamNonBlocking); const unsigned last = N; hipMemcpyAsync((void*)door, &last, 4, hipMemcpyHostToDevice, s2); hipStreamSynchronize(s2); hipStreamDestroy(s2); }
	const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
	hipStreamSynchronize(s);
	printf("%s: %.2f us per round trip\n", what, us);
	hipHostFree(answer); hipStreamDestroy(s);
	return us;
}

struct Find { hsa_agent_t gpu, cpu; bool have_gpu, have_cpu; hsa_amd_memory_pool_t pool[8]; uint32_t pflags[8]; int n_pool; };
static hsa_status_t on_agent(hsa_agent_t a, void* data)
{
	Find* f = (Find*)data; hsa_device_type_t t; hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t);
	if (t == HSA_DEVICE_TYPE_GPU && !f->have_gpu) { f->gpu = a; f->have_gpu = true; }
	if (t == HSA_DEVICE_TYPE_CPU && !f->have_cpu) { f->cpu = a; f->have_cpu = true; }
	return HSA_STATUS_SUCCESS;
}
static hsa_status_t on_pool(hsa_amd_memory_pool_t p, void* data)
{
	Find* f = (Find*)data; hsa_amd_segment_t seg; hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg);
	if (seg != HSA_AMD_SEGMENT_GLOBAL) return HSA_STATUS_SUCCESS;
	uint32_t flags = 0; hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &flags);
	bool alloc = false; hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_RUNTIME_ALLOC_ALLOWED, &alloc);
	hsa_amd_memory_pool_access_t acc; hsa_amd_agent_memory_pool_get_info(f->cpu, p, HSA_AMD_AGENT_MEMORY_POOL_INFO_ACCESS, &acc);
	printf("  gpu pool: flags 0x%x (fine %d coarse %d) alloc %d, cpu access %d (0 never, 1 allowed by default, 2 disallowed by default)\n", flags, !!(flags & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_FINE_GRAINED), !!(flags & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_COARSE_GRAINED), (int)alloc, (int)acc);
	if (alloc && acc != HSA_AMD_MEMORY_POOL_ACCESS_NEVER_ALLOWED && f->n_pool < 8) { f->pflags[f->n_pool] = flags; f->pool[f->n_pool++] = p; }
	return HSA_STATUS_SUCCESS;
}

int main()
{
	hipSetDevice(0);
	{   // baseline: the door in host-mapped memory (what the ray server does today)
		unsigned* h = nullptr; hipHostMalloc((void**)&h, 64, hipHostMallocMapped); *h = 0;
		unsigned* d = nullptr; hipHostGetDevicePointer((void**)&d, h, 0);
		// (host writes through h, the kernel polls d)
		unsigned* answer = nullptr; hipHostMalloc((void**)&answer, 64, hipHostMallocMapped); *answer = 0;
		unsigned* d_answer = nullptr; hipHostGetDevicePointer((void**)&d_answer, answer, 0);
		const unsigned N = 20000; hipStream_t s; hipStreamCreate(&s);
		hipLaunchKernelGGL(echo, dim3(1), dim3(1), 0, s, (volatile unsigned*)d, (volatile unsigned*)d_answer, N);
		const auto t0 = std::chrono::steady_clock::now();
		for (unsigned i = 1; i <= N; ++i) { __atomic_store_n(h, i, __ATOMIC_RELEASE); while (__atomic_load_n(answer, __ATOMIC_ACQUIRE) != i) { } }
		printf("door in host memory: %.2f us per round trip\n", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N);
		hipStreamSynchronize(s);
	}
	{   // probe 1
		void* p = nullptr;
		hipError_t e = hipExtMallocWithFlags(&p, 4096, hipDeviceMallocFinegrained);
		printf("hipExtMallocWithFlags(fine-grained): %s\n", hipGetErrorString(e));
		if (e == hipSuccess) {
			hipMemset(p, 0, 4096); hipDeviceSynchronize();
			signal(SIGSEGV, on_segv); signal(SIGBUS, on_segv);
			if (sigsetjmp(jb, 1) == 0) { *(volatile unsigned*)p = 0u; printf("  host store to it: ok\n"); run((volatile unsigned*)p, "door in fine-grained device memory (hip)", true); }
			else printf("  host store to it: fault\n");
			signal(SIGSEGV, SIG_DFL); signal(SIGBUS, SIG_DFL);
		}
	}
	{   // probe 2
		Find f = {}; hsa_init();
		hsa_iterate_agents(on_agent, &f);
		if (f.have_gpu && f.have_cpu) {
			hsa_amd_agent_iterate_memory_pools(f.gpu, on_pool, &f);
			for (int k = 0; k < f.n_pool; ++k) {
				void* p = nullptr;
				hsa_status_t st = hsa_amd_memory_pool_allocate(f.pool[k], 4096, 0, &p);
				printf("pool flags 0x%x: allocate %d\n", f.pflags[k], (int)st);
				if (st != HSA_STATUS_SUCCESS) continue;
				hsa_agent_t both[2] = { f.cpu, f.gpu };
				st = hsa_amd_agents_allow_access(2, both, nullptr, p);
				printf("  allow access to cpu + gpu: %d\n", (int)st);
				if (st != HSA_STATUS_SUCCESS) continue;
				hipMemset(p, 0, 4096); hipDeviceSynchronize();
				signal(SIGSEGV, on_segv); signal(SIGBUS, on_segv);
				if (sigsetjmp(jb, 1) == 0) { *(volatile unsigned*)p = 0u; printf("  host store: ok\n"); char nm[96]; snprintf(nm, sizeof(nm), "door in device memory, hsa pool flags 0x%x", f.pflags[k]); run((volatile unsigned*)p, nm, true); }
				else printf("  host store: fault\n");
			}

		}
	}
	return 0;
}
