// Growable buffers with a STABLE address: reserve virtual address space once, map physical memory at the tail as the buffer
// grows (`VMMTensor` / `HostVMMTensor` of the reference, corelib/dynamicemb/src/vmm_tensor.cu:30-585, the storage of
// Device / HostExtendableBuffer, dynamicemb/extendable_tensor.py:86-175).
//   device flavour: hipMemAddressReserve + hipMemCreate / hipMemMap / hipMemSetAccess per chunk (HBM);
//   host flavour:   one anonymous mmap reservation (PROT_NONE), chunks committed with mprotect, faulted in and registered
//                   with hipHostRegister(Mapped) -- pinned and addressable by the kernels through the same pointer.
// What a table gains from the stable address: the flat value buffer of a table that grows by rehash keeps its base pointer
// (row addresses = base + slot * row_bytes stay computable from one number), and growing costs no copy of the old rows.
#include "common.h"
#include <stdlib.h>
#include "../../include/recsys_amd.h"
#include <sys/mman.h>
#include <unistd.h>
#include <vector>

namespace {

struct Vmm {
  bool host = false;
  int device = 0;
  char* base = nullptr;
  size_t reserved = 0, mapped = 0, gran = 0;
  std::vector<hipMemGenericAllocationHandle_t> handles;   // device chunks
  std::vector<std::pair<char*, size_t>> registered;       // host chunks
};

bool hip_ok(hipError_t e, const char* what) {
  if (e == hipSuccess) return true;
  char buf[200];
  snprintf(buf, sizeof(buf), "%s: %s", what, hipGetErrorString(e));
  mi355_set_error(buf);
  return false;
}

int map_more(Vmm* v, size_t new_bytes) {
  if (new_bytes <= v->mapped) return MI355_OK;
  size_t want = (new_bytes + v->gran - 1) / v->gran * v->gran;
  if (want > v->reserved) { mi355_set_error("vmm: extend beyond the reserved address range"); return MI355_EINVAL; }
  const size_t add = want - v->mapped;
  // page tables change below: nothing may be in flight on the buffer (measured: a copy still running into the mapped part
  // while the next chunk's access is set is lost).  Growth is rare; a device-wide sync is the simple, safe fence.
  if (!hip_ok(hipDeviceSynchronize(), "hipDeviceSynchronize")) return MI355_ELAUNCH;
  if (!v->host) {
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = v->device;
    hipMemGenericAllocationHandle_t h;
    if (!hip_ok(hipMemCreate(&h, add, &prop, 0), "hipMemCreate")) return MI355_ELAUNCH;
    if (!hip_ok(hipMemMap(v->base + v->mapped, add, 0, h, 0), "hipMemMap")) { hipMemRelease(h); return MI355_ELAUNCH; }
    hipMemAccessDesc acc{};
    acc.location.type = hipMemLocationTypeDevice;
    acc.location.id = v->device;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    // (some ROCm builds only accept an access change over a range that starts at the reservation: fall back to the whole
    // mapped prefix, which re-states the access of the older chunks)
    if (hipMemSetAccess(v->base + v->mapped, add, &acc, 1) != hipSuccess) {
      (void)hipGetLastError();
      if (!hip_ok(hipMemSetAccess(v->base, v->mapped + add, &acc, 1), "hipMemSetAccess")) return MI355_ELAUNCH;
    }
    v->handles.push_back(h);
    if (!hip_ok(hipMemset(v->base + v->mapped, 0, add), "hipMemset")) return MI355_ELAUNCH;
    // ... and the fill must have LANDED before anybody writes rows: hipMemset returns before the blit has run, and with RCCL's
    // queues in the process it was seen to run behind the first forward's first-touch row stores of a fresh module, wiping some of
    // them (round 5: tools/runs/diag_growth.py -- keys found, rows zero).  Mapping is rare: drain the device.
    if (!hip_ok(hipDeviceSynchronize(), "hipDeviceSynchronize")) return MI355_ELAUNCH;
  } else {
    char* p = v->base + v->mapped;
    if (mprotect(p, add, PROT_READ | PROT_WRITE) != 0) { mi355_set_error("vmm: mprotect failed"); return MI355_ELAUNCH; }
    const size_t page = (size_t)sysconf(_SC_PAGESIZE);
    for (size_t o = 0; o < add; o += page) p[o] = 0;        // fault the pages in (anonymous memory reads as zero)
    if (!hip_ok(hipHostRegister(p, add, hipHostRegisterMapped | hipHostRegisterPortable), "hipHostRegister")) return MI355_ELAUNCH;
    void* dp = nullptr;
    if (!hip_ok(hipHostGetDevicePointer(&dp, p, 0), "hipHostGetDevicePointer")) return MI355_ELAUNCH;
    if (dp != (void*)p) { mi355_set_error("vmm: registered host memory is not identity mapped on this system"); return MI355_ELAUNCH; }
    v->registered.emplace_back(p, add);
  }
  v->mapped = want;
  return MI355_OK;
}

}  // namespace

extern "C" {

int mi355_vmm_create(int64_t reserve_bytes, int64_t initial_bytes, int device, int host, void** handle_out) {
  MI355_CHECK_ARG(reserve_bytes > 0 && initial_bytes >= 0 && initial_bytes <= reserve_bytes && handle_out, "vmm: bad sizes");
  Vmm* v = new Vmm();
  v->host = host != 0;
  v->device = device;
  if (!v->host) {
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    size_t g = 0;
    if (!hip_ok(hipMemGetAllocationGranularity(&g, &prop, hipMemAllocationGranularityRecommended), "hipMemGetAllocationGranularity") || g == 0) {
      delete v; return MI355_ELAUNCH;
    }
    v->gran = g;
    v->reserved = ((size_t)reserve_bytes + g - 1) / g * g;
    void* p = nullptr;
    if (!hip_ok(hipMemAddressReserve(&p, v->reserved, g, nullptr, 0), "hipMemAddressReserve")) { delete v; return MI355_ELAUNCH; }
    v->base = (char*)p;
  } else {
    v->gran = 2u << 20;
    v->reserved = ((size_t)reserve_bytes + v->gran - 1) / v->gran * v->gran;
    void* p = mmap(nullptr, v->reserved, PROT_NONE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (p == MAP_FAILED) { mi355_set_error("vmm: mmap reservation failed"); delete v; return MI355_ELAUNCH; }
    v->base = (char*)p;
  }
  if (initial_bytes > 0) {
    const int rc = map_more(v, (size_t)initial_bytes);
    if (rc != MI355_OK) { mi355_vmm_destroy(v); return rc; }
  }
  *handle_out = v;
  return MI355_OK;
}

int mi355_vmm_extend(void* handle, int64_t new_total_bytes) {
  MI355_CHECK_ARG(handle && new_total_bytes >= 0, "vmm: bad arguments");
  return map_more((Vmm*)handle, (size_t)new_total_bytes);
}

void* mi355_vmm_data(void* handle) { return handle ? ((Vmm*)handle)->base : nullptr; }
int64_t mi355_vmm_mapped_bytes(void* handle) { return handle ? (int64_t)((Vmm*)handle)->mapped : 0; }
int64_t mi355_vmm_reserved_bytes(void* handle) { return handle ? (int64_t)((Vmm*)handle)->reserved : 0; }

int mi355_vmm_destroy(void* handle) {
  if (!handle) return MI355_OK;
  Vmm* v = (Vmm*)handle;
  // Page-table changes race with work in flight (see the sync in front of every mapping); a buffer may be released by a finalizer
  // while another module's kernels run.  Unmapping is rare: drain the device first.
  (void)hipDeviceSynchronize();
  if (!v->host) {
    if (v->mapped) hipMemUnmap(v->base, v->mapped);
    for (auto h : v->handles) hipMemRelease(h);
    // The address range is NOT handed back (MI355_VMM_FREE_VA=1 does): a later reservation that lands on a freed range was seen
    // to lose first-touch row stores of its first kernel -- keys found, rows zero, for a third of the rows, only when another
    // extendable buffer had been destroyed just before (round 5, tools/runs/diag_growth.py; stale translations of the old mapping
    // is the only reading that fits).  Virtual address space is not a scarce resource; the physical chunks are released above.
    constexpr bool free_va = false;   // (address ranges are never handed back: DESIGN.md, VMM value buffers)
    if (v->base && free_va) hipMemAddressFree(v->base, v->reserved);
  } else {
    for (auto& r : v->registered) hipHostUnregister(r.first);
    if (v->base) munmap(v->base, v->reserved);
  }
  delete v;
  return MI355_OK;
}

}  // extern "C"
