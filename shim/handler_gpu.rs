//! shim/handler_gpu.rs -- the reference-side binding of libfanlin_gpu.so (include/fanlin_gpu.h, ABI version 3).
//!
//! Add this file to fanlin-rs as `src/gpu.rs` (`mod gpu;` in src/main.rs:14-18) and replace the `image`-crate calls of
//! `State::process_image` (src/handler.rs:221-255, 274-278) as INTEGRATION.md section 3 shows.  Nothing else in the
//! reference changes.  No Rust toolchain exists in the environment this repository is built in, so this file has never
//! been compiled there; `tests/test_abi.py` checks that its #[repr(C)] structs list the header's fields in the header's
//! order, and `tests/c_client.c` exercises the same entry points from C.
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)] pub struct FlImage { data: *mut u8, capacity: u64, width: u32, height: u32, channels: u32, flags: u32, bytes: u64 }
#[repr(C)] #[derive(Default)]
pub struct FlParams { has_dims: u32, w: u32, h: u32, fill_r: u8, fill_g: u8, fill_b: u8, crop: u8,
                      blur_sigma: f32, grayscale: u8, inverse: u8, quality: u8, front_end: u8,
                      orientation: u8, filter: u8 /* 0 Lanczos3, 1 Nearest (GIF frames) */, reserved: [u8; 2] }
#[repr(C)] #[derive(Default)]
pub struct FlPlan { src_w: u32, src_h: u32, mid_c: u32, resampled: u32, resized_w: u32, resized_h: u32, crop_x: u32, crop_y: u32,
                    letterboxed: u32, place_x: u32, place_y: u32, out_w: u32, out_h: u32, out_c: u32,
                    plane_w: u32, plane_h: u32, chroma_w: u32, chroma_h: u32, pixel_bytes: u64, out_bytes: u64, max_out_bytes: u64 }
#[repr(C)] #[derive(Default)]
pub struct FlConfig { device: i32, max_batch: u32, flush_timeout_us: u32, profile: u32, queue_lanes: u32, n_devices: u32,
                      use_embedded_profile: u32 /* config `use_embedded_profile`, handler.rs:19 */, decode_threads: u32 /* 0 = the CPUs the process may use */, devices: [i32; 8] }

pub const FE_NONE: u8 = 0;
pub const FE_JFIF444: u8 = 1;
pub const FE_WEBP420: u8 = 2;
pub const FE_JPEG: u8 = 3;
pub const FILTER_NEAREST: u8 = 1;
const IMG_HAS_ALPHA: u32 = 2;
const CMYK_INPUT_YCCK: u32 = 1;

extern "C" {
    fn flgpu_create(cfg: *const FlConfig, status: *mut c_int) -> *mut c_void;
    fn flgpu_destroy(ctx: *mut c_void);
    fn flgpu_plan_output(p: *const FlParams, sw: u32, sh: u32, sc: u32, plan: *mut FlPlan) -> c_int;
    fn flgpu_transform(ctx: *mut c_void, src: *const FlImage, p: *const FlParams, dst: *mut FlImage) -> c_int;
    fn flgpu_set_cmyk_profile(ctx: *mut c_void, icc: *const u8, n: u64) -> c_int;
    fn flgpu_cmyk_to_rgb(ctx: *mut c_void, cmyk: *const u8, n_pixels: u64, rgb: *mut u8,
                         embedded_icc: *const u8, icc_len: u64, flags: u32) -> c_int;
    fn flgpu_strerror(status: c_int) -> *const c_char;
    fn flgpu_abi_version() -> u32;
}

/// Owned by handler::State next to `cmyk2rgb` (src/handler.rs:14-21); created once at boot (src/main.rs:74-76).
/// The context is internally synchronised: concurrent tokio workers call `transform` and are packed into shared
/// kernel launches by the library's request queue (wrap the call in `spawn_blocking`, it blocks until the result is back).
pub struct Gpu(*mut c_void);
unsafe impl Send for Gpu {}
unsafe impl Sync for Gpu {}

/// What `Gpu::transform` hands back.
pub enum Outcome {
    /// the library ran the request: geometry + the bytes it produced (pixels, planes or a JFIF stream)
    Device { plan: FlPlan, bytes: Vec<u8>, has_alpha: bool },
    /// a pixel layout the device path does not take (16-bit / float `DynamicImage`s from 16-bit PNGs,
    /// src/handler.rs:219): the caller keeps the reference's own CPU code for THIS request, exactly as before
    KeepCpuPath,
}

impl Gpu {
    /// `devices`: HIP ordinals of the node's GPUs (`&[0]` for one); all of them serve the one shared State,
    /// as all tokio workers share one `Arc<State>` (src/main.rs:108-112).
    pub fn new(max_clients: u32, devices: &[i32], use_embedded_profile: bool) -> Result<Self, String> {
        assert_eq!(unsafe { flgpu_abi_version() }, 3, "libfanlin_gpu.so / shim mismatch");
        let mut cfg = FlConfig { device: devices.first().copied().unwrap_or(-1), max_batch: max_clients.max(1),
                                 flush_timeout_us: 200, use_embedded_profile: use_embedded_profile as u32, ..Default::default() };
        if devices.len() > 1 {
            cfg.n_devices = devices.len().min(8) as u32;
            for (k, d) in devices.iter().take(8).enumerate() { cfg.devices[k] = *d; }
        }
        let mut st = 0;
        let p = unsafe { flgpu_create(&cfg, &mut st) };
        if p.is_null() { Err(err(st)) } else { Ok(Gpu(p)) }
    }

    /// main.rs:74-76 `create_cmyk_to_rgb_converter(path)` -> `gpu.set_cmyk_profile(&std::fs::read(path)?)`
    pub fn set_cmyk_profile(&self, icc: &[u8]) -> Result<(), Box<dyn std::error::Error>> {
        check(unsafe { flgpu_set_cmyk_profile(self.0, icc.as_ptr(), icc.len() as u64) })
    }

    /// handler.rs:421-466 after `decoder.decode()`: raw CMYK / YCCK bytes -> RGB8 (lines 423-438 + 469-493).
    pub fn cmyk_to_rgb(&self, raw: &[u8], embedded_icc: Option<&[u8]>, ycck: bool) -> Option<Vec<u8>> {
        let (p, n) = embedded_icc.map_or((std::ptr::null(), 0), |d| (d.as_ptr(), d.len() as u64));
        let mut buf = vec![0u8; raw.len() / 4 * 3];
        let flags = if ycck { CMYK_INPUT_YCCK } else { 0 };
        let rc = unsafe { flgpu_cmyk_to_rgb(self.0, raw.as_ptr(), (raw.len() / 4) as u64, buf.as_mut_ptr(), p, n, flags) };
        if rc != 0 { None } else { Some(buf) } // same as every `?` in the original
    }

    /// Replaces handler.rs:221-255 (+ the colour front end / the whole JPEG encoder when `front_end != 0`).
    /// `orientation`: `decoder.orientation()?.to_exif()` (1..8) read at handler.rs:206; `nearest`: GIF frames (338, 340).
    pub fn transform(&self, img: &image::DynamicImage, q: &crate::query::Query, orientation: u8, front_end: u8, nearest: bool)
        -> Result<Outcome, Box<dyn std::error::Error>>
    {
        use image::DynamicImage::*;
        // 8-bit layouts go to the device as they are; anything else stays on the reference's CPU path for this request
        let (bytes, c): (&[u8], u32) = match img {
            ImageLuma8(b) => (b.as_raw(), 1), ImageLumaA8(b) => (b.as_raw(), 2),
            ImageRgb8(b) => (b.as_raw(), 3), ImageRgba8(b) => (b.as_raw(), 4),
            _ => return Ok(Outcome::KeepCpuPath),
        };
        let (r, g, b) = q.fill_color();
        let mut p = FlParams { fill_r: r, fill_g: g, fill_b: b, crop: q.cropping() as u8, blur_sigma: q.blur(),
                               grayscale: q.grayscale() as u8, inverse: q.inverse() as u8, quality: q.quality(),
                               front_end, orientation, filter: if nearest { FILTER_NEAREST } else { 0 }, ..Default::default() };
        if let Some((w, h)) = q.dimensions() { p.has_dims = 1; p.w = w; p.h = h; }
        let mut plan = FlPlan::default();
        check(unsafe { flgpu_plan_output(&p, img.width(), img.height(), c, &mut plan) })?;
        // allocated by Rust: no cross-allocator frees.  max_out_bytes == out_bytes except for FE_JPEG, where it is the
        // worst case of the format: with it the call cannot fail for lack of room, as `encode_image` into a Vec cannot.
        let mut out = vec![0u8; plan.max_out_bytes as usize];
        let src = FlImage { data: bytes.as_ptr() as *mut u8, capacity: bytes.len() as u64,
                            width: img.width(), height: img.height(), channels: c, flags: 0, bytes: 0 };
        let mut dst = FlImage { data: out.as_mut_ptr(), capacity: out.len() as u64, width: 0, height: 0, channels: 0, flags: 0, bytes: 0 };
        check(unsafe { flgpu_transform(self.0, &src, &p, &mut dst) })?;
        out.truncate(dst.bytes as usize); // pixels / planes: == out_bytes; JPEG: the stream length
        Ok(Outcome::Device { plan, bytes: out, has_alpha: dst.flags & IMG_HAS_ALPHA != 0 })
    }

    /// The pixels of `Outcome::Device` (front_end 0) as the `DynamicImage` the rest of process_image expects.
    pub fn into_dynamic(plan: &FlPlan, pixels: Vec<u8>) -> image::DynamicImage {
        use image::{DynamicImage, ImageBuffer};
        match plan.out_c {
            1 => DynamicImage::ImageLuma8(ImageBuffer::from_raw(plan.out_w, plan.out_h, pixels).unwrap()),
            2 => DynamicImage::ImageLumaA8(ImageBuffer::from_raw(plan.out_w, plan.out_h, pixels).unwrap()),
            3 => DynamicImage::ImageRgb8(ImageBuffer::from_raw(plan.out_w, plan.out_h, pixels).unwrap()),
            _ => DynamicImage::ImageRgba8(ImageBuffer::from_raw(plan.out_w, plan.out_h, pixels).unwrap()),
        }
    }
}
impl Drop for Gpu { fn drop(&mut self) { unsafe { flgpu_destroy(self.0) } } }

fn err(st: c_int) -> String { unsafe { std::ffi::CStr::from_ptr(flgpu_strerror(st)) }.to_string_lossy().into_owned() }
fn check(st: c_int) -> Result<(), Box<dyn std::error::Error>> { if st == 0 { Ok(()) } else { Err(err(st).into()) } }

// ---- how process_image uses it (src/handler.rs:221-255 become) -------------------------------------------------------
//
//     let orientation = decoder.orientation()?;                                     // line 206, kept
//     let mut img = DynamicImage::from_decoder(decoder)?;                            // line 220, kept
//     match self.gpu.transform(&img, params, orientation.to_exif(), gpu::FE_NONE, false)? {
//         gpu::Outcome::Device { plan, bytes, .. } => img = gpu::Gpu::into_dynamic(&plan, bytes),
//         gpu::Outcome::KeepCpuPath => {                                             // 16-bit / float layouts: lines 221-255 as they are
//             img.apply_orientation(orientation);
//             if params.grayscale() { img = img.grayscale(); } else if params.inverse() { img.invert(); }
//             /* ... the reference's own resize / overlay / blur, unchanged ... */
//         }
//     }
//
// and, for JPEG sources that stay JPEG, the whole `ImageFormat::Jpeg` arm (lines 274-278):
//
//     if let gpu::Outcome::Device { bytes, .. } = self.gpu.transform(&img, params, orientation.to_exif(), gpu::FE_JPEG, false)? {
//         return Ok((format.to_mime_type(), bytes));
//     }
