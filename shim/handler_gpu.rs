//! shim/handler_gpu.rs -- the reference-side binding of libfanlin_gpu.so (include/fanlin_gpu.h, ABI version 4).
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

/// flgpu_jpeg_info: what `ImageReader::with_guessed_format` + `JpegDecoder::new` + `decoder.orientation()` learn from the
/// header (src/handler.rs:192-206), for the decision "file bytes to the device, or the reference's own decoder".
#[repr(C)] #[derive(Default, Clone, Copy)]
pub struct FlJpegInfo { width: u32, height: u32, components: u32, channels: u32, progressive: u32, restart_interval: u32,
                        h_max: u32, v_max: u32, exif_orientation: u32, supported: u32, adobe_transform: u32, has_icc_profile: u32 }

pub const FE_NONE: u8 = 0;
pub const FE_JFIF444: u8 = 1;
pub const FE_WEBP420: u8 = 2;
pub const FE_JPEG: u8 = 3;
pub const FILTER_NEAREST: u8 = 1;
const IMG_HAS_ALPHA: u32 = 2;
const IMG_JPEG_SOURCE: u32 = 16; // FlImage.flags of a SOURCE: `data` holds the JPEG FILE (capacity = its length), not pixels
pub const ACCEPT_WEBP: u32 = 1;   // content::Format bits (src/content.rs:12-48)
pub const ACCEPT_AVIF: u32 = 2;
pub const RESULT_AS_IS: c_int = 0;        // flgpu_result_kind
pub const RESULT_JPEG_STREAM: c_int = 1;
pub const RESULT_WEBP_PLANES: c_int = 2;
pub const RESULT_PIXELS: c_int = 3;
const ERR_UNSUPPORTED: c_int = 2; // FLGPU_ERR_UNSUPPORTED: a stream the device decoder does not cover
const CMYK_INPUT_YCCK: u32 = 1;

extern "C" {
    fn flgpu_create(cfg: *const FlConfig, status: *mut c_int) -> *mut c_void;
    fn flgpu_destroy(ctx: *mut c_void);
    fn flgpu_plan_output(p: *const FlParams, sw: u32, sh: u32, sc: u32, plan: *mut FlPlan) -> c_int;
    fn flgpu_transform(ctx: *mut c_void, src: *const FlImage, p: *const FlParams, dst: *mut FlImage) -> c_int;
    fn flgpu_set_cmyk_profile(ctx: *mut c_void, icc: *const u8, n: u64) -> c_int;
    fn flgpu_cmyk_to_rgb(ctx: *mut c_void, cmyk: *const u8, n_pixels: u64, rgb: *mut u8,
                         embedded_icc: *const u8, icc_len: u64, flags: u32) -> c_int;
    fn flgpu_jpeg_info_of(jpeg: *const u8, n: u64, info: *mut FlJpegInfo) -> c_int;
    fn flgpu_process_jpeg_plan(jpeg: *const u8, n: u64, query: *const c_char, accept: u32, plan: *mut FlPlan, kind: *mut c_int) -> c_int;
    fn flgpu_process_jpeg(ctx: *mut c_void, jpeg: *const u8, n: u64, query: *const c_char, accept: u32,
                          dst: *mut FlImage, plan: *mut FlPlan, kind: *mut c_int, out_format: *mut c_int) -> c_int;
    fn flgpu_transform_batch(ctx: *mut c_void, n: usize, srcs: *const FlImage, ps: *const FlParams, dsts: *mut FlImage) -> c_int;
    fn flgpu_strerror(status: c_int) -> *const c_char;
    fn flgpu_abi_version() -> u32;
    fn flgpu_debug_set(ctx: *mut c_void, key: *const c_char, value: i64) -> c_int;
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
        assert_eq!(unsafe { flgpu_abi_version() }, 6, "libfanlin_gpu.so / shim mismatch");
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

    /// A test / experiment switch of the context ("no_mfma", "host_huffman", ...; include/fanlin_gpu.h).  The library reads the process
    /// environment only inside `flgpu_create`: with one `Arc<State>` shared by all workers (src/main.rs:108-112) nothing may call
    /// getenv while another thread may call setenv.
    pub fn debug_set(&self, key: &str, value: i64) -> Result<(), Box<dyn std::error::Error>> {
        let k = std::ffi::CString::new(key)?;
        check(unsafe { flgpu_debug_set(self.0, k.as_ptr(), value) })
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

    /// Header of a JPEG file, or None if the bytes are no JPEG (then `with_guessed_format` decides as before, handler.rs:192-204).
    pub fn jpeg_info(original: &[u8]) -> Option<FlJpegInfo> {
        let mut info = FlJpegInfo::default();
        if unsafe { flgpu_jpeg_info_of(original.as_ptr(), original.len() as u64, &mut info) } == 0 { Some(info) } else { None }
    }

    /// `process_image` for a JPEG input FROM THE FILE BYTES ON (src/handler.rs:192-308 in one call): header, EXIF orientation,
    /// the raw query string, size gate, `as_is`, container negotiation, then Huffman decoding on this thread and IDCT + colour
    /// + CMYK fix + pixel pipeline + encode in one device pass.  `query`: the URL's query string as axum received it.
    /// Ok(None): a stream the device decoder does not cover (arithmetic coding, 12-bit, lossless, hierarchical) -- decode with
    /// the reference's own decoder and use `transform`.  Ok(Some((kind, out_format, plan, bytes))): `kind` says what `bytes` is --
    /// RESULT_AS_IS (return `original` untouched, handler.rs:198-201), RESULT_JPEG_STREAM (the finished image/jpeg body),
    /// RESULT_WEBP_PLANES (Y | U | V for WebPEncode, handler.rs:295-297), RESULT_PIXELS (pixels for the crate's PNG / AVIF /
    /// lossless-WebP encoder; `out_format` = the negotiated container).
    pub fn process_jpeg(&self, original: &[u8], query: &str, accept: u32)
        -> Result<Option<(c_int, c_int, FlPlan, Vec<u8>)>, Box<dyn std::error::Error>>
    {
        let q = std::ffi::CString::new(query)?;
        let (mut plan, mut kind, mut fmt) = (FlPlan::default(), 0 as c_int, 0 as c_int);
        let rc = unsafe { flgpu_process_jpeg_plan(original.as_ptr(), original.len() as u64, q.as_ptr(), accept, &mut plan, &mut kind) };
        if rc == ERR_UNSUPPORTED { return Ok(None); }
        check(rc)?;
        if kind == RESULT_AS_IS { return Ok(Some((kind, 0, plan, Vec::new()))); }
        let mut out = vec![0u8; plan.max_out_bytes as usize];
        let mut dst = FlImage { data: out.as_mut_ptr(), capacity: out.len() as u64, width: 0, height: 0, channels: 0, flags: 0, bytes: 0 };
        let rc = unsafe { flgpu_process_jpeg(self.0, original.as_ptr(), original.len() as u64, q.as_ptr(), accept, &mut dst, &mut plan, &mut kind, &mut fmt) };
        if rc == ERR_UNSUPPORTED { return Ok(None); }
        check(rc)?;
        out.truncate(dst.bytes as usize);
        Ok(Some((kind, fmt, plan, out)))
    }

    /// The same decode + pipeline with parameters already taken from a `Query` (no query string at hand): the SOURCE of `transform`
    /// is the file -- `FlImage.flags = IMG_JPEG_SOURCE`, `width / height / channels` from `jpeg_info`.  The EXIF orientation is
    /// read from the file by the library when `orientation` is 0.
    pub fn transform_jpeg_file(&self, original: &[u8], info: &FlJpegInfo, q: &crate::query::Query, front_end: u8)
        -> Result<Outcome, Box<dyn std::error::Error>>
    {
        if info.supported == 0 { return Ok(Outcome::KeepCpuPath); }
        let (r, g, b) = q.fill_color();
        let mut p = FlParams { fill_r: r, fill_g: g, fill_b: b, crop: q.cropping() as u8, blur_sigma: q.blur(),
                               grayscale: q.grayscale() as u8, inverse: q.inverse() as u8, quality: q.quality(),
                               front_end, orientation: info.exif_orientation as u8, ..Default::default() };
        if let Some((w, h)) = q.dimensions() { p.has_dims = 1; p.w = w; p.h = h; }
        let mut plan = FlPlan::default();
        check(unsafe { flgpu_plan_output(&p, info.width, info.height, info.channels, &mut plan) })?;
        let mut out = vec![0u8; plan.max_out_bytes as usize];
        let src = FlImage { data: original.as_ptr() as *mut u8, capacity: original.len() as u64,
                            width: info.width, height: info.height, channels: info.channels, flags: IMG_JPEG_SOURCE, bytes: 0 };
        let mut dst = FlImage { data: out.as_mut_ptr(), capacity: out.len() as u64, width: 0, height: 0, channels: 0, flags: 0, bytes: 0 };
        check(unsafe { flgpu_transform(self.0, &src, &p, &mut dst) })?;
        out.truncate(dst.bytes as usize);
        Ok(Outcome::Device { plan, bytes: out, has_alpha: dst.flags & IMG_HAS_ALPHA != 0 })
    }

    /// `process_gif` (src/handler.rs:311-366): all frames of an animation in ONE batch -- per frame grayscale / invert, Nearest
    /// resize and letterbox (lines 327-353), `filter = FILTER_NEAREST`.  Frames are Rgba8 buffers of one size; the results come
    /// back in frame order for the GIF encoder (lines 355-363).
    pub fn transform_gif_frames(&self, frames: &[image::RgbaImage], q: &crate::query::Query) -> Result<(FlPlan, Vec<Vec<u8>>), Box<dyn std::error::Error>> {
        let (r, g, b) = q.fill_color();
        let mut p = FlParams { fill_r: r, fill_g: g, fill_b: b, crop: q.cropping() as u8, grayscale: q.grayscale() as u8,
                               inverse: q.inverse() as u8, filter: FILTER_NEAREST, ..Default::default() };
        if let Some((w, h)) = q.dimensions() { p.has_dims = 1; p.w = w; p.h = h; }
        let mut plan = FlPlan::default();
        let (fw, fh) = frames.first().map_or((0, 0), |f| (f.width(), f.height()));
        check(unsafe { flgpu_plan_output(&p, fw, fh, 4, &mut plan) })?;
        let mut outs: Vec<Vec<u8>> = frames.iter().map(|_| vec![0u8; plan.out_bytes as usize]).collect();
        let srcs: Vec<FlImage> = frames.iter().map(|f| FlImage { data: f.as_raw().as_ptr() as *mut u8, capacity: f.as_raw().len() as u64,
                                                                  width: f.width(), height: f.height(), channels: 4, flags: 0, bytes: 0 }).collect();
        let mut dsts: Vec<FlImage> = outs.iter_mut().map(|o| FlImage { data: o.as_mut_ptr(), capacity: o.len() as u64, width: 0, height: 0, channels: 0, flags: 0, bytes: 0 }).collect();
        let ps: Vec<FlParams> = frames.iter().map(|_| FlParams { ..p_clone(&p) }).collect();
        check(unsafe { flgpu_transform_batch(self.0, frames.len(), srcs.as_ptr(), ps.as_ptr(), dsts.as_mut_ptr()) })?;
        Ok((plan, outs))
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

fn p_clone(p: &FlParams) -> FlParams {
    FlParams { has_dims: p.has_dims, w: p.w, h: p.h, fill_r: p.fill_r, fill_g: p.fill_g, fill_b: p.fill_b, crop: p.crop, blur_sigma: p.blur_sigma,
               grayscale: p.grayscale, inverse: p.inverse, quality: p.quality, front_end: p.front_end, orientation: p.orientation, filter: p.filter, reserved: [0; 2] }
}
fn err(st: c_int) -> String { unsafe { std::ffi::CStr::from_ptr(flgpu_strerror(st)) }.to_string_lossy().into_owned() }
fn check(st: c_int) -> Result<(), Box<dyn std::error::Error>> { if st == 0 { Ok(()) } else { Err(err(st).into()) } }

// ---- how process_image uses it ---------------------------------------------------------------------------------------------
// JPEG inputs (src/handler.rs:192-220 + everything below them): hand the FILE over when the header says the device decoder
// covers it -- entropy decoding on this worker thread, everything else on the device, one call:
//
//     if let Some(info) = gpu::Gpu::jpeg_info(original) {                            // instead of with_guessed_format for JPEGs
//         if info.supported == 1 {
//             let accept = (content.webp_accepted() as u32) * gpu::ACCEPT_WEBP | (content.avif_accepted() as u32) * gpu::ACCEPT_AVIF;
//             if let Some((kind, out_format, plan, bytes)) = self.gpu.process_jpeg(original, raw_query, accept)? {
//                 match kind {
//                     gpu::RESULT_AS_IS => return Ok((ImageFormat::Jpeg.to_mime_type(), original.to_vec())),      // lines 198-201
//                     gpu::RESULT_JPEG_STREAM => return Ok((ImageFormat::Jpeg.to_mime_type(), bytes)),            // lines 274-278
//                     gpu::RESULT_WEBP_PLANES => { /* WebPEncode on the Y | U | V planes, lines 295-297 */ }
//                     _ => { let img = gpu::Gpu::into_dynamic(&plan, bytes); /* PNG / AVIF / lossless WebP arm for `out_format` */ }
//                 }
//             }                                                                       // None: fall through to the decoder below
//         }
//     }
//
// Every other input, and JPEGs the device decoder does not cover (src/handler.rs:221-255 become):
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
