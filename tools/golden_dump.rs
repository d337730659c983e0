//! tools/golden_dump.rs -- turns "parity unpinned" into pinned wherever `cargo` exists.
//!
//! Runs the reference's own `image 0.25.6` calls (src/handler.rs:221-278, in the reference's order) on raw inputs
//! exported from tests/golden/oracle_ref.npz by `tools/compare_golden.py export DIR`, and writes what the crate
//! produces next to them; `tools/compare_golden.py compare DIR` then diffs those files against the committed oracle
//! outputs (`__ref` arrays: <= 1 LSB expected for resample / blur, byte-identical for everything else and for the
//! JPEG streams).  Never compiled in this repository's build environment (no Rust toolchain there).
//!
//!   cd tools/golden_dump && cargo run --release -- ../../gpurun_out/golden_cases
//!
//! Each case is a line of DIR/cases.txt:
//!   name h w c  key=value ...      keys: w h crop fill=r,g,b grayscale inverse blur jpeg=quality orientation=1..8
use image::{imageops, imageops::FilterType, DynamicImage, GrayAlphaImage, GrayImage, ImageBuffer, Rgba, RgbImage, RgbaImage};
use std::{fs, path::Path};

fn load(dir: &Path, name: &str, h: u32, w: u32, c: u32) -> DynamicImage {
    let raw = fs::read(dir.join(format!("{name}.in.raw"))).expect("input");
    match c {
        1 => DynamicImage::ImageLuma8(GrayImage::from_raw(w, h, raw).unwrap()),
        2 => DynamicImage::ImageLumaA8(GrayAlphaImage::from_raw(w, h, raw).unwrap()),
        3 => DynamicImage::ImageRgb8(RgbImage::from_raw(w, h, raw).unwrap()),
        _ => DynamicImage::ImageRgba8(RgbaImage::from_raw(w, h, raw).unwrap()),
    }
}

fn main() {
    let dir = std::env::args().nth(1).expect("usage: golden_dump DIR");
    let dir = Path::new(&dir);
    for line in fs::read_to_string(dir.join("cases.txt")).unwrap().lines() {
        let f: Vec<&str> = line.split_whitespace().collect();
        if f.len() < 4 || f[0].starts_with('#') { continue; }
        let (name, h, w, c) = (f[0], f[1].parse().unwrap(), f[2].parse().unwrap(), f[3].parse().unwrap());
        let (mut nw, mut nh, mut crop, mut gray, mut inv, mut blur, mut jpeg, mut orient) = (None, None, false, false, false, None, None, 1u8);
        let mut fill = [32u8, 32, 32];
        for kv in &f[4..] {
            let (k, v) = kv.split_once('=').unwrap_or((kv, "true"));
            match k {
                "w" => nw = v.parse::<u32>().ok(), "h" => nh = v.parse::<u32>().ok(),
                "crop" => crop = v == "true", "grayscale" => gray = v == "true", "inverse" => inv = v == "true",
                "blur" => blur = v.parse::<f32>().ok(), "jpeg" => jpeg = v.parse::<u8>().ok(),
                "orientation" => orient = v.parse().unwrap(),
                "fill" => { let p: Vec<u8> = v.split(',').map(|s| s.parse().unwrap()).collect(); fill = [p[0], p[1], p[2]]; }
                _ => panic!("unknown key {k}"),
            }
        }
        let mut img = load(dir, name, h, w, c);
        // src/handler.rs:221-223
        if orient != 1 { img.apply_orientation(image::metadata::Orientation::from_exif(orient).unwrap()); }
        // :224-228
        if gray { img = img.grayscale(); } else if inv { img.invert(); }
        // :229-249
        if let (Some(w2), Some(h2)) = (nw, nh) {
            if (w2, h2) != (img.width(), img.height()) {
                img = if crop { img.resize_to_fill(w2, h2, FilterType::Lanczos3) } else { img.resize(w2, h2, FilterType::Lanczos3) };
            }
            if w2 > img.width() || h2 > img.height() {
                let mut bg = ImageBuffer::from_pixel(w2, h2, Rgba([fill[0], fill[1], fill[2], 255]));
                let (x, y) = ((w2.abs_diff(img.width()) / 2) as i64, (h2.abs_diff(img.height()) / 2) as i64);
                imageops::overlay(&mut bg, &img, x, y);
                img = DynamicImage::ImageRgba8(bg);
            }
        }
        // :250-255 (sigma already clamped to 10..=20 by Query::blur, src/query.rs:59-62)
        if let Some(s) = blur { img = img.blur(s); }
        fs::write(dir.join(format!("{name}.crate.raw")), img.as_bytes()).unwrap();
        fs::write(dir.join(format!("{name}.crate.shape")), format!("{} {} {}\n", img.height(), img.width(), img.color().channel_count())).unwrap();
        // :274-278
        if let Some(q) = jpeg {
            let mut buf = Vec::new();
            image::codecs::jpeg::JpegEncoder::new_with_quality(&mut buf, q.clamp(1, 100)).encode_image(&img).unwrap();
            fs::write(dir.join(format!("{name}.crate.jpg")), buf).unwrap();
        }
        println!("{name}: {}x{}x{}", img.width(), img.height(), img.color().channel_count());
    }
}
