//! `src/gpu.rs` for razn-v/zkir: a second backend next to `circom.rs`, bound to libzkir_hip.so.
//!
//! Nothing in `ast.rs`, `generator.rs`, `circom.rs` or `rng.rs` changes.  Same shape as
//! `CircomTranspiler` (circom.rs:4-15, 412): a struct that borrows `&Circuit`, one method.
//! NOT COMPILED where this repository is built (no rustc in the image); it is the binding a
//! maintainer would add, written against include/zkir_hip.h.
//!
//! build.rs:  println!("cargo:rustc-link-search=native=<repo>/zkir_amd");
//!            println!("cargo:rustc-link-lib=dylib=zkir_hip");
#![allow(dead_code)]

use crate::ast::*;
use std::ffi::{c_void, CStr, CString};
use std::os::raw::c_char;

#[repr(C)]
#[derive(Default, Clone, Copy)]
pub struct TapeHdr {
    pub magic: u32, pub version: u32, pub n_inputs: u32, pub n_outputs: u32, pub n_consts: u32,
    pub n_ops: u32, pub n_gslots: u32, pub n_lslots: u32, pub n_constraints: u32, pub n_mul_ops: u32,
    pub flags: u32, pub strand_ops: [u32; 8], pub n_xslots: u32, pub n_strands: u32, pub n_scale_ops: u32,
}

/// zkir_lower_options of the header; `LowerOptions::default()` is NOT the library's default:
/// fill it with zkir_lower_options_default, then set `strands = 4` (or 8, for batches up to 2^14 vectors) to ask
/// for a strand tape (that many wavefronts per 64 vectors; circuits it does not fit come back as plain tapes);
/// a campaign, bound by the host side, sets `reorder = 0`.
#[repr(C)]
#[derive(Default, Clone, Copy)]
pub struct LowerOptions {
    pub fold_constants: u32, pub n_lslots: u32, pub reuse_slots: u32, pub max_ops: u32, pub bnot_variant: u32,
    pub use_acc: u32, pub max_iterations: u32, pub window_lifetime: u32, pub strands: u32, pub max_xslots: u32,
    pub reorder: u32, pub window_cache: u32, pub order_ops: u32, pub scaled_values: u32,
}

#[link(name = "zkir_hip")]
extern "C" {
    fn zkir_host_last_error() -> *const c_char;
    fn zkir_circuit_parse(text: *const c_char, out: *mut *mut c_void) -> i32;
    fn zkir_circuit_free(c: *mut c_void);
    fn zkir_lower_options_default(o: *mut LowerOptions) -> i32;
    fn zkir_lower(c: *const c_void, opts: *const LowerOptions, out: *mut *mut c_void, class: *mut i32) -> i32;
    /// generate + lower for a batch of circuits on the library's own host threads (cfgs: *const zkir_generator_config or null)
    fn zkir_generate_lower_many(seeds: *const u64, cfgs: *const c_void, n: u32, opts: *const LowerOptions, threads: u32,
                                tapes: *mut *mut c_void, classes: *mut i32) -> i32;
    fn zkir_tape_header(t: *const c_void, hdr: *mut TapeHdr) -> i32;
    fn zkir_tape_blob(t: *const c_void, blob: *mut *const c_void, bytes: *mut usize) -> i32;
    fn zkir_tape_free(t: *mut c_void);
    fn zkir_ctx_create(device: i32, out: *mut *mut c_void) -> i32;
    fn zkir_ctx_destroy(ctx: *mut c_void) -> i32;
    fn zkir_last_error(ctx: *const c_void) -> *const c_char;
    fn zkir_tape_upload(ctx: *mut c_void, hdr: *const TapeHdr, blob: *const c_void, bytes: usize, id: *mut u32) -> i32;
    fn zkir_tape_release(ctx: *mut c_void, id: u32) -> i32;
    // a launch's worth of tapes in one copy (campaigns); ids come back in `out[0..n)`
    fn zkir_tape_upload_many(ctx: *mut c_void, n: u32, hdrs: *const TapeHdr, blobs: *const *const c_void,
                             bytes: *const usize, out: *mut u32) -> i32;
    // the same for tape handles this library lowered (zkir_lower, zkir_generate_lower_many): not validated again
    fn zkir_tape_upload_handles(ctx: *mut c_void, n: u32, tapes: *const *const c_void, out: *mut u32) -> i32;
    // thread-seconds the library's host threads have spent generating and lowering since it was loaded
    fn zkir_host_phase_seconds(generate_s: *mut f64, lower_s: *mut f64) -> i32;
    // zkir_eval_batch on device pointers and a stream; `_noalloc`: the same on the library's fixed scratch
    // array (no hipMalloc in the call: for hosts that run under HIP graph capture)
    fn zkir_eval_batch_dev(ctx: *mut c_void, tape: u32, inputs_soa_dev: *const u32, n: u64, outputs_soa_dev: *mut u32,
                           result_dev: *mut u64, stream: *mut c_void) -> i32;
    fn zkir_eval_batch_dev_noalloc(ctx: *mut c_void, tape: u32, inputs_soa_dev: *const u32, n: u64,
                                   outputs_soa_dev: *mut u32, result_dev: *mut u64, stream: *mut c_void) -> i32;
    fn zkir_eval_multi(ctx: *mut c_void, tapes: *const u32, n_tapes: u32, inputs: *const u32, n_per_tape: u64,
                       per_tape_unsat: *mut u64) -> i32;
    fn zkir_eval_batch(ctx: *mut c_void, tape: u32, inputs_soa: *const u32, n: u64, outputs_soa: *mut u32,
                       unsat_count: *mut u64, first_unsat: *mut u32) -> i32;
    // many tapes, one launch, device buffers; per_tape_faults_dev (may be null) names the circuit that divided by zero
    fn zkir_eval_multi_dev_faults(ctx: *mut c_void, tapes: *const u32, n_tapes: u32, inputs_dev: *const u32, n_per_tape: u64,
                                  per_tape_unsat_dev: *mut u64, per_tape_faults_dev: *mut u64, result_dev: *mut u64,
                                  stream: *mut c_void) -> i32;
    // the campaign's exchange step for one process that drives several GPUs (RCCL inside): counters summed,
    // divergent circuit ids united; host memory in and out
    fn zkir_reducer_create(devices: *const i32, n_devices: u32, n_counters: u32, id_capacity: u32, out: *mut *mut c_void) -> i32;
    fn zkir_reducer_reduce(r: *mut c_void, counters: *const u64, ids: *const u32, n_ids: *const u32, totals: *mut u64,
                           all_ids: *mut u32, n_all_ids: *mut u32) -> i32;
    fn zkir_reducer_destroy(r: *mut c_void) -> i32;
}

/// Why `lower` refused a circuit (ZKIR_LOWER_* of the header); circom is expected to refuse the
/// same circuits, so a campaign counts "both refuse" as agreement.
#[derive(Debug)]
pub enum GpuError { Lower { class: i32, message: String }, Library { code: i32, message: String } }

fn host_error() -> String { unsafe { CStr::from_ptr(zkir_host_last_error()).to_string_lossy().into_owned() } }

// ---- the text form a Circuit crosses the ABI in (grammar: zkir_amd/csrc/host/sexpr.hpp) ----
fn var_ref(r: &VarRef) -> String {
    match r.idx { Some(i) => format!("({} {})", r.id, i), None => format!("{}", r.id) }
}

fn expr(e: &Expr) -> String {
    let bin = |op: &str, l: &Expr, r: &Expr| format!("({} {} {})", op, expr(l), expr(r));
    match e {
        Expr::Var(r) => format!("(var {})", var_ref(r)),
        Expr::Constant(f) => format!("(const \"{}\")", f),
        Expr::Add(l, r) => bin("add", l, r), Expr::Sub(l, r) => bin("sub", l, r), Expr::Mul(l, r) => bin("mul", l, r),
        Expr::Power(l, r) => bin("pow", l, r), Expr::Div(l, r) => bin("div", l, r),
        Expr::IntDiv(l, r) => bin("idiv", l, r), Expr::Rem(l, r) => bin("rem", l, r),
        Expr::LessThan(l, r) => bin("lt", l, r), Expr::LessThanEq(l, r) => bin("le", l, r),
        Expr::GreaterThan(l, r) => bin("gt", l, r), Expr::GreaterThanEq(l, r) => bin("ge", l, r),
        Expr::Equal(l, r) => bin("eq", l, r), Expr::And(l, r) => bin("and", l, r), Expr::Or(l, r) => bin("or", l, r),
        Expr::BitAnd(l, r) => bin("band", l, r), Expr::BitOr(l, r) => bin("bor", l, r),
        Expr::BitNot(x) => format!("(bnot {})", expr(x)), Expr::BitXor(l, r) => bin("bxor", l, r),
        Expr::BitRShift(l, r) => bin("shr", l, r), Expr::BitLShift(l, r) => bin("shl", l, r),
    }
}

fn block(tag: &str, instrs: &[Instruction]) -> String {
    format!("({} {})", tag, instrs.iter().map(instr).collect::<Vec<_>>().join(" "))
}

fn instr(i: &Instruction) -> String {
    match i {
        Instruction::VarDecl(v, None) => format!("(decl {})", v.var_ref.id),
        Instruction::VarDecl(v, Some(e)) => format!("(decl {} {})", v.var_ref.id, expr(e)),
        Instruction::Assign(r, e) => format!("(assign {} {})", var_ref(r), expr(e)),
        Instruction::ArrayAssign(r, idx, e) => format!("(aassign {} {} {})", var_ref(r), expr(idx), expr(e)),
        Instruction::Constraint(r, e) => format!("(constrain {} {})", var_ref(r), expr(e)),
        Instruction::If { cond, then_branch, else_branch } => match else_branch {
            Some(eb) => format!("(if {} {} {})", expr(cond), block("then", then_branch), block("else", eb)),
            None => format!("(if {} {})", expr(cond), block("then", then_branch)),
        },
        Instruction::For { init, cond, step, body } =>
            format!("(for {} {} {} {})", instr(init), expr(cond), instr(step), block("body", body)),
        Instruction::While { cond, body } => format!("(while {} {})", expr(cond), block("body", body)),
    }
}

pub fn to_sexpr(c: &Circuit) -> String {
    let vars: Vec<String> = c.variables.iter().map(|v| {
        let ty = match v.var_type { VariableType::Field => "field".to_string(), VariableType::Array(n) => format!("(array {})", n) };
        let role = match v.role {
            VariableRole::Local => "local",
            VariableRole::Signal(SignalType::Input) => "input",
            VariableRole::Signal(SignalType::Output) => "output",
        };
        let init: Vec<&str> = v.initialized.iter().map(|&b| if b { "1" } else { "0" }).collect();
        format!("(var {} \"{}\" {} {} (init {}))", v.var_ref.id, v.name, ty, role, init.join(" "))
    }).collect();
    format!("(circuit (vars {}) (instrs {}))", vars.join(" "),
            c.instructions.iter().map(instr).collect::<Vec<_>>().join(" "))
}

/// Evaluates one circuit for batches of input vectors on an MI355X.
pub struct GpuEvaluator<'a> {
    pub circuit: &'a Circuit,
    ctx: *mut c_void,
    tape: u32,
    pub hdr: TapeHdr,
}

impl<'a> GpuEvaluator<'a> {
    pub fn new(circuit: &'a Circuit, device: i32) -> Result<Self, GpuError> {
        unsafe {
            let text = CString::new(to_sexpr(circuit)).unwrap();
            let mut c: *mut c_void = std::ptr::null_mut();
            let rc = zkir_circuit_parse(text.as_ptr(), &mut c);
            if rc != 0 { return Err(GpuError::Library { code: rc, message: host_error() }); }
            let (mut t, mut class) = (std::ptr::null_mut(), 0i32);
            let rc = zkir_lower(c, std::ptr::null(), &mut t, &mut class);
            zkir_circuit_free(c);
            if rc != 0 { return Err(GpuError::Lower { class, message: host_error() }); }
            let mut hdr = TapeHdr::default();
            let (mut blob, mut bytes) = (std::ptr::null(), 0usize);
            zkir_tape_header(t, &mut hdr);
            zkir_tape_blob(t, &mut blob, &mut bytes);
            let mut ctx: *mut c_void = std::ptr::null_mut();
            let rc = zkir_ctx_create(device, &mut ctx);
            if rc != 0 { zkir_tape_free(t); return Err(GpuError::Library { code: rc, message: "no gfx950 device".into() }); }
            let mut id = 0u32;
            let rc = zkir_tape_upload(ctx, &hdr, blob, bytes, &mut id);
            zkir_tape_free(t);
            if rc != 0 {
                let message = CStr::from_ptr(zkir_last_error(ctx)).to_string_lossy().into_owned();
                zkir_ctx_destroy(ctx);
                return Err(GpuError::Library { code: rc, message });
            }
            Ok(Self { circuit, ctx, tape: id, hdr })
        }
    }

    /// inputs_soa: [n_inputs][8][n] u32 limbs (canonical, little endian); outputs_soa: [n_outputs][8][n].
    /// Returns (unsatisfied (vector, constraint) pairs, first such pair).
    pub fn eval_batch(&mut self, inputs_soa: &[u32], n: u64, outputs_soa: &mut [u32])
                      -> Result<(u64, Option<(u32, u32)>), GpuError> {
        assert_eq!(inputs_soa.len() as u64, self.hdr.n_inputs as u64 * 8 * n);
        assert_eq!(outputs_soa.len() as u64, self.hdr.n_outputs as u64 * 8 * n);
        let (mut unsat, mut first) = (0u64, [0u32; 2]);
        let rc = unsafe {
            zkir_eval_batch(self.ctx, self.tape, inputs_soa.as_ptr(), n, outputs_soa.as_mut_ptr(), &mut unsat, first.as_mut_ptr())
        };
        if rc != 0 {
            let message = unsafe { CStr::from_ptr(zkir_last_error(self.ctx)).to_string_lossy().into_owned() };
            return Err(GpuError::Library { code: rc, message });
        }
        Ok((unsat, if first[0] == u32::MAX { None } else { Some((first[0], first[1])) }))
    }
}

impl<'a> Drop for GpuEvaluator<'a> {
    fn drop(&mut self) {
        unsafe { zkir_tape_release(self.ctx, self.tape); zkir_ctx_destroy(self.ctx); }
    }
}
