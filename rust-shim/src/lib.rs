//! jubjub-hip: batched Jubjub arithmetic on AMD MI355X (gfx950) behind the `jubjub` crate's own types.
//!
//! A thin, safe layer over `libjubjub_hip.so` (C ABI: `include/jubjub_hip.h`; raw declarations: `ffi.rs`, generated from that header by
//! `tools/gen_rust_ffi.py` and held to it by `tests/test_rust_shim_signatures.py`).  It marshals through the crate's PUBLIC byte APIs only
//! (`Fr::to_bytes / from_bytes`, `Fq::to_bytes / from_bytes`, `AffinePoint::{get_u, get_v, from_raw_unchecked, to_bytes}`), so it builds next
//! to an unmodified `jubjub` 0.10.  The build image of the GPU library has no Rust toolchain: this crate has not been compiled there; what IS
//! checked mechanically is every `extern "C"` signature against the header.
//!
//! Reference interface replaced, by entry point: see INTEGRATION.md section 2 (`/root/reference/src/lib.rs`, `src/fr.rs` line numbers).
use jubjub::{AffinePoint, ExtendedPoint, Fq, Fr, SubgroupPoint};
use group::{cofactor::CofactorGroup, Curve};
use std::os::raw::{c_int, c_void};

pub mod ffi;
use ffi::*;


pub const ZIP216: u32 = 1; pub const TORSION_FREE: u32 = 2; pub const NOT_SMALL_ORDER: u32 = 4; pub const CLEAR_COFACTOR: u32 = 8;

/// A byte buffer in page-locked host memory (jj_host_alloc) for callers that multiply batch after batch: allocated ONCE (page-locking
/// a gigabyte takes longer than the call it serves), reused by every call (`PinnedBatch` below), it lets the copies run straight from
/// and to the caller's memory: 0.89-0.91 of the device-resident rate at 2^24 fixed-base units (profiles/r4_pcie_inclusive.txt).
/// One-shot callers just pass `Vec<u8>`s: the library moves pageable memory through its own page-locked staging buffers with a few copy
/// threads (within 2-3 % of the page-locked rate, and a freshly allocated result vector costs only its page faults).
pub struct HostBuf { p: *mut u8, len: usize }
impl HostBuf {
    pub fn new(len: usize) -> Self {
        let mut p: *mut c_void = std::ptr::null_mut();
        assert_eq!(unsafe { jj_host_alloc(len.max(1), &mut p) }, 0);
        HostBuf { p: p as *mut u8, len }
    }
    pub fn as_slice(&self) -> &[u8] { unsafe { std::slice::from_raw_parts(self.p, self.len) } }
    pub fn as_mut_slice(&mut self) -> &mut [u8] { unsafe { std::slice::from_raw_parts_mut(self.p, self.len) } }
    pub fn as_ptr(&self) -> *const c_void { self.p as _ }
    pub fn as_mut_ptr(&mut self) -> *mut c_void { self.p as _ }
}
impl Drop for HostBuf { fn drop(&mut self) { unsafe { jj_host_free(self.p as _); } } }
unsafe impl Send for HostBuf {}

/// an MSM in flight (jj_msm_begin): finished exactly once by `GpuBatch::msm_finish`
pub struct MsmJob { job: *mut JjMsmJob, _keep: (Vec<u8>, Vec<u8>) }

fn write_points(p: &mut [u8], points: &[AffinePoint]) {
    for (i, q) in points.iter().enumerate() {
        p[64 * i..64 * i + 32].copy_from_slice(&q.get_u().to_bytes());        // src/lib.rs:630-637
        p[64 * i + 32..64 * i + 64].copy_from_slice(&q.get_v().to_bytes());
    }
}
fn write_scalars(s: &mut [u8], scalars: &[Fr]) {
    for (i, k) in scalars.iter().enumerate() { s[32 * i..32 * i + 32].copy_from_slice(&k.to_bytes()); }   // src/fr.rs:296-308
}
fn put_points(points: &[AffinePoint]) -> Vec<u8> { let mut p = vec![0u8; 64 * points.len()]; write_points(&mut p, points); p }
fn put_scalars(scalars: &[Fr]) -> Vec<u8> { let mut s = vec![0u8; 32 * scalars.len()]; write_scalars(&mut s, scalars); s }

/// Marshalling buffers of a caller that multiplies batch after batch: page-locked once, reused by every call.
pub struct PinnedBatch { cap: usize, s: HostBuf, p: HostBuf, out: HostBuf }
impl PinnedBatch {
    pub fn new(cap: usize) -> Self { PinnedBatch { cap, s: HostBuf::new(32 * cap), p: HostBuf::new(64 * cap), out: HostBuf::new(64 * cap) } }
}
fn get_point(b: &[u8]) -> AffinePoint {
    let u = Fq::from_bytes(b[..32].try_into().unwrap()).unwrap();                 // the library only emits canonical encodings
    let v = Fq::from_bytes(b[32..64].try_into().unwrap()).unwrap();
    AffinePoint::from_raw_unchecked(u, v)                                         // src/lib.rs:662-664
}

/// A page-locked result buffer out of the context's pool (jj_result_acquire); goes back to the pool when dropped.
pub struct Pooled<'a> { gpu: &'a Gpu, p: *mut u8, len: usize }
impl Pooled<'_> {
    pub fn bytes(&self) -> &[u8] { unsafe { std::slice::from_raw_parts(self.p, self.len) } }
}
impl Drop for Pooled<'_> { fn drop(&mut self) { unsafe { jj_result_release(self.gpu.0, self.p as _); } } }

pub struct Gpu(*mut JjCtx);
unsafe impl Send for Gpu {}    // the library serialises the entry points of one context on a lock
unsafe impl Sync for Gpu {}

impl Gpu {
    pub fn new(device: i32) -> Option<Self> {
        let mut p = std::ptr::null_mut();
        if unsafe { jj_ctx_create(device, &mut p) } == 0 { Some(Gpu(p)) } else { None }   // JJ_ERR_NODEVICE: no gfx950, no CPU fallback
    }

    /// Per-context tuning by key (jj_ctx_set_option; the library reads no environment variable).  No key touches the timing discipline.
    pub fn set_option(&self, key: &str, value: i64) -> bool {
        let k = std::ffi::CString::new(key).unwrap();
        unsafe { jj_ctx_set_option(self.0, k.as_ptr(), value) == 0 }
    }

    /// Batched `points[i] * scalars[i]`  (reference: `Mul<&Fr> for &ExtendedPoint`, src/lib.rs:873-879).  CONSTANT-TIME like the reference's
    /// ladder (`conditional_select`, src/lib.rs:334-343, 357-379): `jj_varbase_mul` has no scalar-dependent address or branch.
    pub fn mul_batch(&self, points: &[AffinePoint], scalars: &[Fr]) -> Vec<AffinePoint> {
        assert_eq!(points.len(), scalars.len());                                  // cf. src/lib.rs:841
        let (n, s, p) = (points.len(), put_scalars(scalars), put_points(points)); // pageable inputs: through the library's staging slots (bounce path)
        let out = self.result(64 * n);                                            // a page-locked buffer from the library's pool: no page faults of a fresh
                                                                                  // Vec inside the call, no registration; a different buffer per call in flight
        assert_eq!(unsafe { jj_varbase_mul(self.0, n, s.as_ptr() as _, p.as_ptr() as _, out.p as _) }, 0);
        out.bytes().chunks_exact(64).map(get_point).collect()                     // the `-> Vec` the caller sees; `out` goes back to the pool when it drops here
    }

    /// Result buffers for the `-> Vec<..>`-shaped functions (reference: `batch_from_bytes` src/lib.rs:541-627, `batch_normalize` 1084-1107): see
    /// `jj_result_acquire` in include/jubjub_hip.h.  Measured (bench.py --host-buffers pooled, 2^24 fixed-base units): 559 M/s against 195 M/s with a fresh `Vec`.
    fn result(&self, bytes: usize) -> Pooled<'_> {
        let mut p = std::ptr::null_mut();
        assert_eq!(unsafe { jj_result_acquire(self.0, bytes, &mut p) }, 0);
        Pooled { gpu: self, p: p as *mut u8, len: bytes }
    }

    /// The same with the caller's page-locked, reused buffers: the copies run straight from and to them.
    pub fn mul_batch_pinned(&self, b: &mut PinnedBatch, points: &[AffinePoint], scalars: &[Fr]) -> Vec<AffinePoint> {
        assert!(points.len() == scalars.len() && points.len() <= b.cap);
        let n = points.len();
        write_scalars(b.s.as_mut_slice(), scalars);
        write_points(b.p.as_mut_slice(), points);
        assert_eq!(unsafe { jj_varbase_mul(self.0, n, b.s.as_ptr(), b.p.as_ptr(), b.out.as_mut_ptr()) }, 0);
        b.out.as_slice()[..64 * n].chunks_exact(64).map(get_point).collect()
    }

    /// One scalar, many bases: the `Wnaf::new().scalar(k)` then `.base(p)` reuse pattern (WnafGroup, src/lib.rs:1318-1336).
    pub fn mul_scalar(&self, k: &Fr, points: &[AffinePoint]) -> Vec<AffinePoint> {
        let (n, p) = (points.len(), put_points(points));
        let mut out = vec![0u8; 64 * n];
        assert_eq!(unsafe { jj_varbase_mul_scalar(self.0, n, k.to_bytes().as_ptr() as _, p.as_ptr() as _, out.as_mut_ptr() as _) }, 0);
        out.chunks_exact(64).map(get_point).collect()
    }

    /// `sum_i points[i] * scalars[i]` (the `Sum` of `p * k`, src/lib.rs:183-193).
    pub fn msm(&self, points: &[AffinePoint], scalars: &[Fr]) -> ExtendedPoint {
        assert_eq!(points.len(), scalars.len());
        let (n, s, p) = (points.len(), put_scalars(scalars), put_points(points));
        let mut out = [0u8; 64];
        assert_eq!(unsafe { jj_msm(self.0, n, s.as_ptr() as _, p.as_ptr() as _, out.as_mut_ptr() as _) }, 0);
        get_point(&out).into()
    }

    /// `AffinePoint::batch_from_bytes` (src/lib.rs:541-627); `flags`: ZIP216 (from_bytes vs from_bytes_pre_zip216_compatibility,
    /// src/lib.rs:469-489), TORSION_FREE (SubgroupPoint::from_bytes, 1427-1429), NOT_SMALL_ORDER (699-705), CLEAR_COFACTOR (722-724).
    pub fn batch_from_bytes(&self, enc: &[[u8; 32]], flags: u32) -> Vec<Option<AffinePoint>> {
        let n = enc.len();
        let (mut out, mut ok) = (vec![0u8; 64 * n], vec![0u8; n]);                // `enc` is already the wire format: passed as it is
        assert_eq!(unsafe { jj_decompress(self.0, n, enc.as_ptr() as _, flags, out.as_mut_ptr() as _, ok.as_mut_ptr()) }, 0);
        (0..n).map(|i| if ok[i] == 1 { Some(get_point(&out[64 * i..64 * i + 64])) } else { None }).collect()   // CtOption -> Option
    }

    /// `SubgroupPoint::from_bytes` for a whole vector (decode + `[r]P == O`, src/lib.rs:1427-1429).
    pub fn subgroup_points_from_bytes(&self, enc: &[[u8; 32]]) -> Vec<Option<SubgroupPoint>> {
        self.batch_from_bytes(enc, ZIP216 | TORSION_FREE).into_iter()
            .map(|p| p.map(|a| ExtendedPoint::from(a).into_subgroup().unwrap())).collect()   // the check already ran on the GPU
    }

    /// `AffinePoint::to_bytes` for a whole vector (src/lib.rs:455-464).
    pub fn to_bytes_batch(&self, points: &[AffinePoint]) -> Vec<[u8; 32]> {
        let (n, p) = (points.len(), put_points(points));
        let mut out = vec![[0u8; 32]; n];
        assert_eq!(unsafe { jj_compress(self.0, n, p.as_ptr() as _, out.as_mut_ptr() as _) }, 0);
        out
    }

    /// `ExtendedPoint::batch_normalize` (src/lib.rs:1084-1107).  NOT BOUND to `jj_batch_normalize` from outside the crate:
    /// `ExtendedPoint`'s five coordinates are private (src/lib.rs:138-145) and the crate exposes no accessor, so a shim that lives
    /// next to the crate cannot hand (U, V, Z, T1, T2) to the GPU; it falls back to the crate's own per-point `to_affine()`.
    /// Inside the crate (a `pub(crate)` accessor, three lines) the entry point takes the 160-byte `(U,V,Z,T1,T2)` records as they are;
    /// that binding is exercised by the C++ mirror (`batch_normalize` in include/jubjub_hip.hpp) and the Python one, not from Rust.
    pub fn batch_normalize(&self, pts: &[ExtendedPoint]) -> Vec<AffinePoint> {
        pts.iter().map(|p| p.to_affine()).collect()
    }

    /// MSM with the host tail of one call overlapping the kernels of the next: `begin` queues the device work, `finish` waits for
    /// that job only (iterator `Sum` of `p * k`, src/lib.rs:183-193 + 873-879).
    pub fn msm_begin(&self, scalars: &[Fr], points: &[AffinePoint]) -> MsmJob {
        assert_eq!(scalars.len(), points.len());
        let (s, p) = (put_scalars(scalars), put_points(points));
        let mut job: *mut JjMsmJob = std::ptr::null_mut();
        assert_eq!(unsafe { jj_msm_begin(self.0, scalars.len(), s.as_ptr() as _, p.as_ptr() as _, &mut job) }, 0);
        MsmJob { job, _keep: (s, p) }                       // host arrays are staged at begin; kept alive until finish anyway
    }
    pub fn msm_finish(&self, j: MsmJob) -> AffinePoint {
        let mut out = [0u8; 64];
        assert_eq!(unsafe { jj_msm_finish(j.job, out.as_mut_ptr() as _) }, 0);
        get_point(&out)
    }

    /// The same product for PUBLIC scalars only (verification keys, public randomisers): the variable-time ladder, whose per-lane window
    /// table is read at digit-dependent addresses -- ~1.6 % faster.  Not a drop-in for `Mul<Fr>` (the reference's `multiply` is constant-time,
    /// src/lib.rs:357-379); it stands where a caller would reach for the group crate's `Wnaf`, which is variable-time by design.
    pub fn multiply_batch_vartime(&self, scalars: &[Fr], points: &[AffinePoint]) -> Vec<AffinePoint> {
        assert_eq!(scalars.len(), points.len());
        let (n, s, p) = (scalars.len(), put_scalars(scalars), put_points(points));
        let mut out = vec![0u8; 64 * n];
        assert_eq!(unsafe { jj_varbase_mul_vartime(self.0, n, s.as_ptr() as _, p.as_ptr() as _, out.as_mut_ptr() as _) }, 0);
        (0..n).map(|i| get_point(&out[64 * i..64 * i + 64])).collect()
    }

    /// `is_torsion_free` (src/lib.rs:709-711) for a whole vector.
    pub fn is_torsion_free_batch(&self, points: &[AffinePoint]) -> Vec<bool> {
        let (n, p) = (points.len(), put_points(points));
        let mut out = vec![0u8; n];
        assert_eq!(unsafe { jj_is_torsion_free(self.0, n, p.as_ptr() as _, out.as_mut_ptr()) }, 0);
        out.into_iter().map(|b| b == 1).collect()
    }

    /// `PrimeFieldBits::to_le_bits` (src/fr.rs:746-773) for a whole vector: 256 bits (one byte each) per scalar.
    pub fn to_le_bits_batch(&self, scalars: &[Fr]) -> Vec<[u8; 256]> {
        let (n, s) = (scalars.len(), put_scalars(scalars));
        let mut out = vec![[0u8; 256]; n];
        assert_eq!(unsafe { jj_fr_to_le_bits(self.0, n, s.as_ptr() as _, out.as_mut_ptr() as _) }, 0);
        out
    }
}
impl Drop for Gpu { fn drop(&mut self) { unsafe { jj_ctx_destroy(self.0); } } }

/// `AffineNielsPoint * Fr` / `multiply_bits` (src/lib.rs:272-310) for one fixed base: the table lives on the device.
pub struct FixedBase<'a> { gpu: &'a Gpu, t: *mut JjTable }
impl<'a> FixedBase<'a> {
    pub fn new(gpu: &'a Gpu, base: &AffinePoint) -> Self {
        let (b, mut t) = (put_points(std::slice::from_ref(base)), std::ptr::null_mut());
        assert_eq!(unsafe { jj_fixedbase_table_create(gpu.0, b.as_ptr() as _, 0, &mut t) }, 0);   // 0: LDS table, constant-time select
        FixedBase { gpu, t }
    }
    pub fn mul_batch(&self, scalars: &[Fr]) -> Vec<AffinePoint> {
        let (n, s) = (scalars.len(), put_scalars(scalars));
        let mut out = vec![0u8; 64 * n];
        assert_eq!(unsafe { jj_fixedbase_mul(self.gpu.0, self.t, n, s.as_ptr() as _, out.as_mut_ptr() as _) }, 0);
        out.chunks_exact(64).map(get_point).collect()
    }
}
impl Drop for FixedBase<'_> { fn drop(&mut self) { unsafe { jj_fixedbase_table_destroy(self.gpu.0, self.t); } } }

/// All GPUs of the node from one process: contiguous shards, one host thread and stream per device.
pub struct Node(*mut JjMulti);
impl Node {
    pub fn new(devices: &[i32]) -> Option<Self> {
        let mut p = std::ptr::null_mut();
        if unsafe { jj_multi_create(devices.as_ptr(), devices.len() as c_int, &mut p) } == 0 { Some(Node(p)) } else { None }
    }
    pub fn mul_batch(&self, points: &[AffinePoint], scalars: &[Fr]) -> Vec<AffinePoint> {
        let (n, s, p) = (points.len(), put_scalars(scalars), put_points(points));
        let mut out = vec![0u8; 64 * n];
        assert_eq!(unsafe { jj_multi_varbase_mul(self.0, n, s.as_ptr() as _, p.as_ptr() as _, out.as_mut_ptr() as _) }, 0);
        out.chunks_exact(64).map(get_point).collect()
    }
    pub fn msm(&self, points: &[AffinePoint], scalars: &[Fr]) -> ExtendedPoint {
        let (n, s, p) = (points.len(), put_scalars(scalars), put_points(points));
        let mut out = [0u8; 64];
        assert_eq!(unsafe { jj_multi_msm(self.0, n, s.as_ptr() as _, p.as_ptr() as _, out.as_mut_ptr() as _) }, 0);
        get_point(&out).into()
    }
}
impl Drop for Node { fn drop(&mut self) { unsafe { jj_multi_destroy(self.0); } } }

/// One process per GPU (the layout `north_star` names): this rank's terms stay resident on its GPU, the records of window sums
/// (8256 bytes per rank) are all-gathered over RCCL / xGMI and every rank runs one host tail.  The application owns the communicator:
/// `comm` is an `ncclComm_t` it made with `ncclCommInitRank` (e.g. through an `rccl-sys` binding; the ncclUniqueId travels over
/// whatever rendezvous the application has -- examples/msm_rccl.cpp uses a file), `all_gather` the address of that library's
/// `ncclAllGather` (or null: looked up in the process, then in librccl.so.1).  examples/msm_rccl.cpp is this sequence in C++, built and
/// run by tests/test_gpu_host_path.py.
impl Gpu {
    pub fn set_comm(&self, comm: *mut c_void, rank: i32, nranks: i32, all_gather: *mut c_void) {
        assert_eq!(unsafe { jj_ctx_set_comm(self.0, comm, rank, nranks, all_gather) }, 0);
    }
    /// every rank passes ITS terms (partition 0) and gets the same sum; a collective: all ranks call it, in the same order
    pub fn msm_all_ranks(&self, my_points: &[AffinePoint], my_scalars: &[Fr]) -> ExtendedPoint {
        let (n, s, p) = (my_points.len(), put_scalars(my_scalars), put_points(my_points));
        let mut out = [0u8; 64];
        assert_eq!(unsafe { jj_msm_allgather(self.0, n, s.as_ptr() as _, p.as_ptr() as _, 0, out.as_mut_ptr() as _) }, 0);
        get_point(&out).into()
    }
    /// the same in two halves for a stream of sums (`msm_finish` above takes the job): with 2-4 jobs in flight the gather, the fold
    /// and the host tail of one sum run beside the kernels of the next; every rank begins the same jobs in the same order
    pub fn msm_all_ranks_begin(&self, my_points: &[AffinePoint], my_scalars: &[Fr]) -> MsmJob {
        let (n, s, p) = (my_points.len(), put_scalars(my_scalars), put_points(my_points));
        let mut job: *mut JjMsmJob = std::ptr::null_mut();
        assert_eq!(unsafe { jj_msm_allgather_begin(self.0, n, s.as_ptr() as _, p.as_ptr() as _, 0, &mut job) }, 0);
        MsmJob { job, _keep: (s, p) }
    }
}
