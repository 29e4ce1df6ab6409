//! src/hip_backend.rs for zlogic/matrix-eyes: the safe layer over hip_ffi.rs that `reconstruction::extract_depth`
//! calls under `#[cfg(feature = "hip")]` in place of `DepthProModelLoader::extract_depth::<B>` + `DepthMap::new::<B>`
//! (reconstruction.rs:155-205).  NOT COMPILED in the build image (no Rust toolchain there); the compiled twin
//! of this file is matrix-eyes_amd/host/matrix_eyes.{hpp,cpp}, function for function.
#![cfg(feature = "hip")]
use crate::hip_ffi as ffi;
use std::ffi::{c_char, c_void, CStr, CString};
use std::ptr::{null, null_mut};

#[derive(Debug)]
pub struct HipError {
    pub code: i32,
    pub message: String,
}
impl std::fmt::Display for HipError {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        write!(f, "HIP back end error {}: {}", self.code, self.message)
    }
}
impl std::error::Error for HipError {}

/// reconstruction::init_device (reconstruction.rs:42-72): one context = one GPU, one stream, resident weights.
pub struct HipDevice {
    ctx: *mut ffi::MeCtx,
    weights_loaded: std::cell::Cell<bool>,
}

impl HipDevice {
    pub fn new(device_id: i32) -> Result<HipDevice, HipError> {
        let mut ctx: *mut ffi::MeCtx = null_mut();
        let rc = unsafe { ffi::me_ctx_create(device_id, ffi::ME_DTYPE_F16, null(), &mut ctx) };
        if rc != ffi::ME_OK {
            return Err(last_error(null(), rc));
        }
        Ok(HipDevice { ctx, weights_loaded: std::cell::Cell::new(false) })
    }

    /// A loop over many images (main.rs called per file of a batch): `.obj` files are written behind the caller by up
    /// to `files_in_flight` host threads while the GPU works on the next image; `flush_outputs` before the files are
    /// read.  0 restores the reference's form (output_mesh returns with the file written).
    pub fn set_write_behind(&self, files_in_flight: i32) -> Result<(), HipError> {
        self.check(unsafe { ffi::me_ctx_set_write_behind(self.ctx, files_in_flight) })
    }

    /// Waits for every file a write-behind `output_mesh` has handed to a host thread; a failed write surfaces here
    /// (or at the next `output_mesh` that has to wait for it) as the OutputError::Io it would have been.
    pub fn flush_outputs(&self) -> Result<(), HipError> {
        self.check(unsafe { ffi::me_output_flush(self.ctx) })
    }

    fn check(&self, rc: i32) -> Result<(), HipError> {
        if rc == ffi::ME_OK {
            Ok(())
        } else {
            Err(last_error(self.ctx, rc))
        }
    }

    /// mod.rs:174-249 load_record, once per device: the library reads the PyTorch archive itself (no key remap,
    /// no transpose adapter: it takes PyTorch names and layouts).
    pub fn ensure_loaded(&self, checkpoint_path: &str) -> Result<(), HipError> {
        if self.weights_loaded.get() {
            return Ok(());
        }
        let path = CString::new(checkpoint_path).map_err(|_| HipError { code: ffi::ME_ERR_BAD_ARG, message: "path".into() })?;
        self.check(unsafe { ffi::me_load_checkpoint_pt(self.ctx, path.as_ptr()) })?;
        self.weights_loaded.set(true);
        Ok(())
    }

    /// mod.rs:251-363 extract_depth on the u8 HWC pixels of SourceImage::load (reconstruction.rs:114): the
    /// normalise / permute of :116-124 runs on the GPU.  f_norm None = FOV head (mod.rs:343-358).
    pub fn extract_depth_u8<PL: crate::depth_pro::ProgressListener>(
        &self, rgb8: &[u8], size: usize, f_norm: Option<f32>, pl: Option<PL>,
    ) -> Result<Vec<f32>, HipError> {
        assert_eq!(rgb8.len(), size * size * 3);
        let mut depth = vec![0f32; size * size];
        let mut boxed = pl;
        unsafe extern "C" fn trampoline<PL: crate::depth_pro::ProgressListener>(user: *mut c_void, pos: f32, msg: *const c_char) {
            let pl = &*(user as *const PL);
            pl.report_status(pos);
            if !msg.is_null() {
                pl.update_message(CStr::from_ptr(msg).to_string_lossy().into_owned());
            }
        }
        if let Some(p) = boxed.as_mut() {
            self.check(unsafe { ffi::me_ctx_set_progress(self.ctx, Some(trampoline::<PL>), p as *mut PL as *mut c_void) })?;
        }
        let fn_ptr = f_norm.as_ref().map_or(null(), |f| f as *const f32);
        let rc = unsafe { ffi::me_extract_depth_u8(self.ctx, rgb8.as_ptr(), 1, fn_ptr, depth.as_mut_ptr(), null_mut()) };
        unsafe { ffi::me_ctx_set_progress(self.ctx, None, null_mut()) };
        self.check(rc)?;
        Ok(depth)
    }
}

impl Drop for HipDevice {
    fn drop(&mut self) {
        unsafe { ffi::me_ctx_destroy(self.ctx) }
    }
}

fn last_error(ctx: *const ffi::MeCtx, code: i32) -> HipError {
    let message = unsafe { CStr::from_ptr(ffi::me_last_error(ctx)) }.to_string_lossy().into_owned();
    HipError { code, message }
}

/// output::DepthMap (output.rs:40-121) over the C ABI: `data` stays on the host like the reference's, the kernels
/// take it from there.
pub struct DepthMap<'d> {
    device: &'d HipDevice,
    pub data: Vec<f32>,
    pub data_width: usize,
    pub data_height: usize,
    pub original_size: (u32, u32),
    range: (f32, f32),
}

impl<'d> DepthMap<'d> {
    /// output.rs:44-75: clamp to [1/250, 1/0.1] + inverse_depth_range
    pub fn from_vec(device: &'d HipDevice, mut data: Vec<f32>, dims: [usize; 2], original_size: (u32, u32)) -> Result<Self, HipError> {
        let (mut mn, mut mx) = (0f32, 0f32);
        device.check(unsafe { ffi::me_depth_clamp_minmax(device.ctx, data.as_mut_ptr(), data.len() as i64, &mut mn, &mut mx) })?;
        Ok(DepthMap { device, data, data_width: dims[0], data_height: dims[1], original_size, range: (mn, mx) })
    }

    /// output.rs:123-131 output_depth_map (before resize_exact + save, which stay in Rust)
    pub fn depth_map_rgb(&self) -> Result<Vec<u8>, HipError> {
        let mut rgb = vec![0u8; self.data.len() * 3];
        self.device.check(unsafe {
            ffi::me_depthmap_rgb(self.device.ctx, self.data.as_ptr(), self.data.len() as i64, self.range.0, self.range.1, rgb.as_mut_ptr())
        })?;
        Ok(rgb)
    }

    /// output.rs:141-193 output_stereogram; `noise` is what :165-171 draws (one [u8; 3] per pixel, row by row)
    pub fn stereogram(&self, out_w: u32, out_h: u32, amplitude: f32, noise: &[u8]) -> Result<Vec<u8>, HipError> {
        assert_eq!(noise.len(), out_w as usize * out_h as usize * 3);
        let mut out = vec![0u8; noise.len()];
        self.device.check(unsafe {
            ffi::me_stereogram(self.device.ctx, self.data.as_ptr(), self.data_width as i32, self.data_height as i32, self.range.0,
                               self.range.1, out_w as i32, out_h as i32, amplitude, noise.as_ptr(), out.as_mut_ptr())
        })?;
        Ok(out)
    }

    /// output.rs:195-261 output_mesh with ObjWriter / PlyWriter (:385-630); vertex_mode = ME_VERTEX_*
    pub fn output_mesh(&self, destination_path: &str, source_path: &str, vertex_mode: i32, vertex_colors: Option<&[u8]>) -> Result<(), HipError> {
        let dst = CString::new(destination_path).unwrap();
        let src = CString::new(source_path).unwrap();
        self.device.check(unsafe {
            ffi::me_output_mesh(self.device.ctx, self.data.as_ptr(), self.data_width as i32, self.data_height as i32,
                                self.original_size.0, self.original_size.1, dst.as_ptr(), src.as_ptr(), vertex_mode,
                                vertex_colors.map_or(null(), |c| c.as_ptr()))
        })
    }
}
