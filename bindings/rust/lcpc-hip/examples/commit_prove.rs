//! `cargo run --release --example commit_prove -- 24`: commit to a random polynomial with 2^24 Ft255 coefficients on the GPU,
//! prove an evaluation, and let the REFERENCE's verifier (lcpc-2d lib.rs:518-527, running `encode` through the GPU encoder)
//! check it -- the shape of lcpc-ligero-pc/src/tests.rs:100-177 with `HipLigeroEncoding` / `HipCommit` in place of
//! `LigeroEncoding` / `LigeroCommit`.
use blake3::Hasher as Blake3;
use ff::Field;
use lcpc_2d::{LcEncoding, LcEvalProof};
use lcpc_hip::{HipCommit, HipLigeroEncoding, HipTranscript};
use lcpc_test_fields::ft255::Ft255;
use std::time::Instant;

fn powers(x: Ft255, n: usize, step: usize) -> Vec<Ft255> {
    let xs = x.pow_vartime(&[step as u64]);
    std::iter::successors(Some(Ft255::one()), |p| Some(*p * xs)).take(n).collect()
}

fn main() {
    let lgl: usize = std::env::args().nth(1).and_then(|s| s.parse().ok()).unwrap_or(20);
    let len = 1usize << lgl;
    let mut rng = rand::thread_rng();
    let coeffs: Vec<Ft255> = std::iter::repeat_with(|| Ft255::random(&mut rng)).take(len).collect();

    let enc = HipLigeroEncoding::<Ft255>::new(len); // twiddle tables and packs go to the device here, outside any timing
    let t0 = Instant::now();
    let comm = HipCommit::commit(&coeffs, &enc).expect("commit");
    let root = comm.get_root();
    println!("commit 2^{}: {:?} (from host memory: PCIe-inclusive)", lgl, t0.elapsed());

    let x = Ft255::random(&mut rng);
    let (n_rows, n_per_row, _) = enc.get_dims(len);
    let inner = powers(x, n_per_row, 1);
    let outer = powers(x, n_rows, n_per_row);

    let mut tr = HipTranscript::new(b"example");
    tr.append_message(b"polycommit", root.as_ref());
    let t0 = Instant::now();
    let proof: LcEvalProof<Blake3, HipLigeroEncoding<Ft255>> = comm.prove(&outer, &enc, &mut tr).expect("prove");
    println!("prove: {:?}, {} bytes", t0.elapsed(), bincode::serialize(&proof).unwrap().len());

    // the reference's own verify, generic over LcEncoding: merlin's transcript on this side
    let mut vtr = merlin::Transcript::new(b"example");
    vtr.append_message(b"polycommit", root.as_ref());
    let eval = proof.verify(root.as_ref(), &outer, &inner, &enc, &mut vtr).expect("verify");
    let direct = coeffs.iter().rev().fold(Ft255::zero(), |acc, c| acc * x + c); // Horner
    assert_eq!(eval, direct);
    println!("verified: p(x) = {:?}", eval);
}
