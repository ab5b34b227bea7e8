# KmerGMAHIP.jl -- Julia shim over libkgma (include/kgma.h): drop-in replacements for the scan engines of
# KmerGMA.jl with the SAME keyword signatures, so `findGenes` / `findGenes_cluster_mode` (src/API.jl:83-94,
# :201-216) can call them instead of `ac_gma_testing!` / `Omn_KmerGMA!`, and `record_KmerGMA!`
# (src/MultiThread/GenomeMiner.jl:8-98) for callers that scan one record at a time.
#
# NOTE: Julia is not installed in the build container nor on the GPU box, so this file has never been
# executed.  It is deliberately thin (marshalling only) and mirrors, call for call, the Python ctypes host
# (kmergma_amd/_lib.py, kmergma_amd/api.py) that IS tested against the same library.
#
# What stays in Julia (exactly the reference's own code): reference preparation (gen_ref_ws_cons /
# cluster_ref_API), re-alignment of hits (BioAlignments.pairalign + cigar_to_UnitRange) and FASTA.Record
# construction.  What moves to the GPU: FASTA parsing + encoding (kgma_genome_from_fasta: the file is mapped
# and goes to the device ONCE; no LongDNA copy of the genome is made on the host) and the per-record body of
# the engines (src/GenomeMiner.jl:32-107, src/OmnGenomeMiner.jl:55-160).  Residues are read back only for the
# hits: `kgma_genome_fetch` serves the alignment segments, `kgma_genome_fetch_batch` the record bodies of all hits at once.

module KmerGMAHIP

using KmerGMA, FASTX, BioSequences, BioAlignments, Mmap

const libkgma = get(ENV, "KGMA_LIB", joinpath(@__DIR__, "..", "libkgma.so"))

struct KgmaHit            # must match kgma_hit in include/kgma.h (64 bytes)
    contig::Int32
    kfv::Int32
    cmi::Int64
    lo::Int64
    hi::Int64
    genome_pos::Int64
    dist::Float64
    D::Int64
    flags::UInt32
    reserved::UInt32
end

const KGMA_MODE_SINGLE = Int32(0)
const KGMA_MODE_OMN = Int32(1)
const KGMA_F_RETURN_DISTS = UInt32(1)
const KGMA_F_CHAIN_REPLAY = UInt32(4)   # ties decided by the reference's running Float64 distance (kgma.h)
const KGMA_E_BADBASE = 4
const KGMA_E_BOUNDS = 5

mutable struct Context
    h::Ptr{Cvoid}
    function Context(device::Integer = 0)
        r = Ref{Ptr{Cvoid}}(C_NULL)
        st = ccall((:kgma_create, libkgma), Cint, (Cint, Ref{Ptr{Cvoid}}), device, r)
        st == 0 || error("kgma_create failed: " * unsafe_string(ccall((:kgma_status_string, libkgma), Cstring, (Cint,), st)))
        ctx = new(r[])
        finalizer(c -> (c.h == C_NULL || ccall((:kgma_destroy, libkgma), Cvoid, (Ptr{Cvoid},), c.h); c.h = C_NULL), ctx)
        return ctx
    end
end

const DEFAULT_CTX = Ref{Union{Nothing, Context}}(nothing)
default_context() = (DEFAULT_CTX[] === nothing && (DEFAULT_CTX[] = Context(0)); DEFAULT_CTX[])

function check(ctx::Context, st::Integer)
    st == 0 && return
    msg = unsafe_string(ccall((:kgma_last_error, libkgma), Cstring, (Ptr{Cvoid},), ctx.h))
    st == KGMA_E_BADBASE && throw(KeyError(msg))        # NUCLEOTIDE_BITS lookup, src/Consts.jl:22-28
    st == KGMA_E_BOUNDS && throw(BoundsError(msg))      # src/OmnGenomeMiner.jl:84-86
    error("libkgma status $st: $msg")
end

# ---- the genome lives on the device; the host keeps a handle ------------------------------------------------
mutable struct DeviceGenome
    ctx::Context
    h::Ptr{Cvoid}
end

# FASTA file -> device (replaces `open(FASTA.Reader, genome_path)` + getSeq, src/GenomeMiner.jl:31-35): the library reads
# the file straight into its pinned staging buffers (kgma_genome_from_fasta_file), strips the line breaks and encodes the
# residues on the GPU
function genome_from_fasta(ctx::Context, genome_path::String)
    g = Ref{Ptr{Cvoid}}(C_NULL)
    check(ctx, ccall((:kgma_genome_from_fasta_file, libkgma), Cint, (Ptr{Cvoid}, Cstring, Ref{Ptr{Cvoid}}), ctx.h, genome_path, g))
    return DeviceGenome(ctx, g[])
end

# one record (record_KmerGMA!): its residues as FASTX holds them
function genome_from_record(ctx::Context, record::FASTA.Record)
    s = Vector{UInt8}(FASTA.sequence(String, record))
    g = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve s begin
        check(ctx, ccall((:kgma_genome_from_host, libkgma), Cint,
                         (Ptr{Cvoid}, Ptr{Ptr{UInt8}}, Ptr{Int64}, Int64, Ref{Ptr{Cvoid}}),
                         ctx.h, [pointer(s)], Int64[length(s)], 1, g))
    end
    return DeviceGenome(ctx, g[])
end

free!(g::DeviceGenome) = (g.h == C_NULL || ccall((:kgma_genome_free, libkgma), Cvoid, (Ptr{Cvoid}, Ptr{Cvoid}), g.ctx.h, g.h); g.h = C_NULL; nothing)

function identifier(g::DeviceGenome, contig::Integer)
    txt = Ref{Cstring}(C_NULL); n = Ref{Int64}(0)
    st = ccall((:kgma_genome_header, libkgma), Cint, (Ptr{Cvoid}, Int64, Ref{Cstring}, Ref{Int64}), g.h, contig, txt, n)
    st == 0 || error("the genome has no FASTA headers")
    hdr = unsafe_string(Ptr{UInt8}(txt[]), n[])
    return String(first(split(hdr, (' ', '\t'); limit = 2)))     # FASTA.identifier: up to the first whitespace
end

# view(seq, lo:hi) as a LongDNA{4} (the residues travel back from the device: hits only)
function subseq(g::DeviceGenome, contig::Integer, lo::Integer, hi::Integer)
    hi < lo && return KmerGMA.Seq("")
    buf = Vector{UInt8}(undef, hi - lo + 1)
    check(g.ctx, ccall((:kgma_genome_fetch, libkgma), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Int64, Ptr{UInt8}),
                       g.ctx.h, g.h, contig, lo, hi - lo + 1, buf))
    return KmerGMA.Seq(String(buf))
end

# view(seq, lo:hi) of every hit in ONE device gather + ONE download (kgma_genome_fetch_batch)
function hit_bodies(g::DeviceGenome, hits)
    n = length(hits)
    n == 0 && return KmerGMA.Seq[]
    contigs = Int64[h.contig for h in hits]
    pos = Int64[h.lo for h in hits]
    lens = Int64[max(h.hi - h.lo + 1, 0) for h in hits]
    buf = Vector{UInt8}(undef, max(sum(lens), 1))
    check(g.ctx, ccall((:kgma_genome_fetch_batch, libkgma), Cint,
                       (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}, Ptr{UInt8}, Int64),
                       g.ctx.h, g.h, n, contigs, pos, lens, buf, length(buf)))
    offs = cumsum(vcat(0, lens))
    return [KmerGMA.Seq(String(buf[offs[i]+1:offs[i+1]])) for i in 1:n]
end

function fetch_hits(ctx::Context)
    n = Ref{Int64}(0)
    check(ctx, ccall((:kgma_get_hits, libkgma), Cint, (Ptr{Cvoid}, Ptr{KgmaHit}, Int64, Ref{Int64}), ctx.h, C_NULL, 0, n))
    hits = Vector{KgmaHit}(undef, n[])
    check(ctx, ccall((:kgma_get_hits, libkgma), Cint, (Ptr{Cvoid}, Ptr{KgmaHit}, Int64, Ref{Int64}), ctx.h, hits, n[], n))
    return hits
end

function fetch_dists!(ctx::Context, kfv::Integer, dist_vec)
    n = Ref{Int64}(0)
    check(ctx, ccall((:kgma_get_dists, libkgma), Cint, (Ptr{Cvoid}, Int32, Ptr{Float64}, Int64, Ref{Int64}), ctx.h, kfv, C_NULL, 0, n))
    buf = Vector{Float64}(undef, n[])
    check(ctx, ccall((:kgma_get_dists, libkgma), Cint, (Ptr{Cvoid}, Int32, Ptr{Float64}, Int64, Ref{Int64}), ctx.h, kfv, buf, n[], n))
    append!(dist_vec, buf)
end

# kmer_dist(seq, KFV, k) (src/Kmers.jl:58-60) of many sequences against one KFV in one device batch:
# what cluster_ref_API (ReferenceGeneration.jl:101) and estimate_optimal_threshold (DistanceTesting.jl:14,27)
# compute one sequence at a time.
function kmer_dist_batch(ctx::Context, seqs::Vector{<:AbstractString}, KFV::Vector{Float64}, k::Int)
    length(KFV) == 4^k || error("the KFV must have 4^k entries")
    text = Vector{UInt8}(join(seqs))
    offsets = Int64[0; cumsum(Int64[ncodeunits(s) for s in seqs])]
    out = Vector{Float64}(undef, length(seqs))
    check(ctx, ccall((:kgma_kmer_dist_batch, libkgma), Cint,
                     (Ptr{Cvoid}, Int32, Ptr{Float64}, Ptr{UInt8}, Ptr{Int64}, Int64, Ptr{Float64}),
                     ctx.h, k, KFV, text, offsets, length(seqs), out))
    return out
end

# alignment callback: the library calls this in reference order with the candidate range and
# expects the aligned range back (Alignment.jl:33-52 / OmnGenomeMiner.jl:130-136).
mutable struct AlignState
    genome::DeviceGenome
    consensus::Vector{KmerGMA.Seq}
    windowsize::Int            # > 0: single engine (view(consensus, 1:windowsize)); 0: cluster engine
    score_model
    keep::Bool
    out::Vector
end

function align_trampoline(user::Ptr{Cvoid}, contig::Int32, kfv::Int32, lo::Int64, hi::Int64, L::Int64,
                          out_lo::Ptr{Int64}, out_hi::Ptr{Int64})::Cvoid
    st = unsafe_pointer_to_objref(user)::AlignState
    segment = subseq(st.genome, contig, lo, hi)                        # view(seq, lo:hi)
    cons = kfv == 0 ? view(st.consensus[1], 1:st.windowsize) : st.consensus[kfv]
    aligned_obj = pairalign(SemiGlobalAlignment(), cons, segment, st.score_model)
    st.keep && push!(st.out, aligned_obj)
    r = cigar_to_UnitRange(aligned_obj)
    unsafe_store!(out_lo, max(1, lo + first(r) - 1))
    unsafe_store!(out_hi, min(lo + last(r) - 1, L))
    return
end

scan_flags(do_return_dists::Bool, float_chain::Bool) =
    (do_return_dists ? KGMA_F_RETURN_DISTS : UInt32(0)) | (float_chain ? KGMA_F_CHAIN_REPLAY : UInt32(0))

# the body shared by ac_gma_testing! and record_KmerGMA!: scan `g` with the single engine and build the records
function single_engine!(ctx::Context, g::DeviceGenome, ident::Function; refVec, consensus_refseq, k, windowsize, thr, buff,
                        do_align, result_align_vec, gap_open_score, gap_extend_score, do_return_dists, dist_vec,
                        do_return_align, get_hit_loci, hit_loci_vec, resultVec, n_refs, with_genome_pos::Bool, float_chain::Bool)
    nref = n_refs === nothing ? C_NULL : Int64[n_refs]
    check(ctx, ccall((:kgma_set_refs, libkgma), Cint,
        (Ptr{Cvoid}, Int32, Int32, Ptr{Float64}, Ptr{Int64}, Ptr{Float64}, Ptr{Int64}),
        ctx.h, k, 1, collect(Float64, refVec), Int64[windowsize], Float64[thr], nref))
    st = AlignState(g, [consensus_refseq], windowsize,
        AffineGapScoreModel(EDNAFULL, gap_open = gap_open_score, gap_extend = gap_extend_score),
        do_return_align, result_align_vec)
    cb = do_align ? @cfunction(align_trampoline, Cvoid,
        (Ptr{Cvoid}, Int32, Int32, Int64, Int64, Int64, Ptr{Int64}, Ptr{Int64})) : C_NULL
    GC.@preserve st begin
        check(ctx, ccall((:kgma_scan, libkgma), Cint,
            (Ptr{Cvoid}, Ptr{Cvoid}, Int32, Int64, Int64, UInt32, Ptr{Cvoid}, Ptr{Cvoid}),
            ctx.h, g.h, KGMA_MODE_SINGLE, buff, 0, scan_flags(do_return_dists, float_chain), cb, pointer_from_objref(st)))
    end
    hits = fetch_hits(ctx)
    bodies = hit_bodies(g, hits)
    for (h, body) in zip(hits, bodies)
        seq_UnitRange = Int(h.lo):Int(h.hi)
        # the reference's record format: src/Alignment.jl:69-80 (append_hit!, do_overlap = false);
        # record_KmerGMA! omits GenomePos (src/MultiThread/GenomeMiner.jl:87-93)
        header = ident(h.contig) *
            " | dist = " * string(round(h.dist, digits = 2)) *
            " | MatchPos = $seq_UnitRange" *
            (with_genome_pos ? " | GenomePos = $(h.genome_pos)" : "") *
            " | Len = " * string(last(seq_UnitRange) - first(seq_UnitRange) + 1)
        push!(resultVec, FASTA.Record(header, body))
        get_hit_loci && push!(hit_loci_vec, h.lo + h.genome_pos)
    end
    do_return_dists && fetch_dists!(ctx, 1, dist_vec)
    return nothing
end

"""
    ac_gma_testing!(; kwargs...)   -- same keywords as KmerGMA.ac_gma_testing! (src/GenomeMiner.jl:4-23)
plus `n_refs` (number of reference sequences averaged into refVec; inferred when omitted) and `float_chain`
(default true: rounding-dependent ties decided by the reference's running Float64 distance, KGMA_F_CHAIN_REPLAY).
"""
function ac_gma_testing!(; genome_path::String, refVec::Vector{Float64}, consensus_refseq::KmerGMA.Seq,
    k::Int64 = 6, windowsize::Int64 = 289, thr::Union{Int64, Float64} = 33.5, buff::Int64 = 50,
    mask::UInt64 = unsigned(4095), Nt_bits = NUCLEOTIDE_BITS, ScaleFactor::Float64 = 1/6,
    do_align::Bool = true, result_align_vec = [], gap_open_score::Int = -69, gap_extend_score::Int = -1,
    do_return_dists::Bool = false, dist_vec = Float64[], do_return_align::Bool = false,
    get_hit_loci::Bool = false, hit_loci_vec::Vector{Int} = Int[],
    resultVec::Vector{FASTA.Record} = FASTA.Record[], n_refs::Union{Nothing, Int} = nothing,
    float_chain::Bool = true, ctx::Context = default_context())

    mask == unsigned(4^k - 1) || error("mask must be 4^k - 1")
    g = genome_from_fasta(ctx, genome_path)
    try
        single_engine!(ctx, g, c -> identifier(g, c); refVec, consensus_refseq, k, windowsize, thr, buff, do_align,
            result_align_vec, gap_open_score, gap_extend_score, do_return_dists, dist_vec, do_return_align,
            get_hit_loci, hit_loci_vec, resultVec, n_refs, with_genome_pos = true, float_chain)
    finally
        free!(g)
    end
    return nothing
end

"""
    record_KmerGMA!(; kwargs...)   -- same keywords as KmerGMA.record_KmerGMA! (src/MultiThread/GenomeMiner.jl:8-23):
one record in, hits pushed to `resultVec_vec[Threads.threadid()]`, headers without GenomePos.  `refVec` may be the
reference's SVector or a plain Vector; `curr_kmer_freq_vec` (the per-thread count buffers) is accepted and unused:
the counts live on the device.  Use one Context per Julia thread (a context is not thread-safe).
"""
function record_KmerGMA!(; record::FASTA.Record, refVec, curr_kmer_freq_vec = nothing, consensus_refseq::KmerGMA.Seq,
    resultVec_vec::Vector{Vector{FASTA.Record}}, k::Int64 = 6, windowsize::Int64 = 289,
    thr::Union{Int64, Float64} = 30, buff::Int64 = 50, mask::UInt64 = unsigned(4095), Nt_bits = NUCLEOTIDE_BITS,
    ScaleFactor::Float64 = 1/6, initial_scale_factor::Float64 = 1/12, do_align::Bool = true,
    score_model::AffineGapScoreModel{Int64} = AffineGapScoreModel(EDNAFULL, gap_open = -69, gap_extend = -1),
    n_refs::Union{Nothing, Int} = nothing, float_chain::Bool = true, ctx::Context = default_context())

    mask == unsigned(4^k - 1) || error("mask must be 4^k - 1")
    g = genome_from_record(ctx, record)
    try
        single_engine!(ctx, g, c -> FASTA.identifier(record); refVec, consensus_refseq, k, windowsize, thr, buff, do_align,
            result_align_vec = [], gap_open_score = score_model.gap_open, gap_extend_score = score_model.gap_extend,
            do_return_dists = false, dist_vec = Float64[], do_return_align = false, get_hit_loci = false,
            hit_loci_vec = Int[], resultVec = resultVec_vec[Threads.threadid()], n_refs, with_genome_pos = false, float_chain)
    finally
        free!(g)
    end
    return nothing
end

"""
    Omn_KmerGMA!(; kwargs...)   -- same keywords as KmerGMA.Omn_KmerGMA! (src/OmnGenomeMiner.jl:7-30)
plus `n_refs::Vector{Int}` (reference count per cluster) and `float_chain`.
"""
function Omn_KmerGMA!(; genome_path::String, refVecs::Vector{Vector{Float64}}, windowsizes::Vector{Int64},
    consensus_seqs::Vector{KmerGMA.Seq}, resultVec::Vector{FASTA.Record}, k::Int64 = 6, ScaleFactor::Real = 1/6,
    mask::UInt64 = unsigned(4095), thr_vec = Float64[35, 31, 38, 34, 27, 27], buff::Int64 = 50,
    Nt_bits = NUCLEOTIDE_BITS, align_hits::Bool = true, align_vec = [], gap_open_score::Int = -200,
    gap_extend_score::Int = -1, genome_pos::Int64 = 0, get_hit_loci::Bool = false,
    hit_loci_vec::Vector{Int} = Int[], get_aligns::Bool = false, do_return_dists::Bool = false,
    dist_vec_vec::Vector{Vector{Float64}} = [Float64[] for _ in 1:6],
    n_refs::Union{Nothing, Vector{Int}} = nothing, float_chain::Bool = true, ctx::Context = default_context())

    m = length(windowsizes)
    refmat = reduce(vcat, refVecs[1:m])                  # m x 4^k, row-major as the C side expects
    nref = n_refs === nothing ? C_NULL : Int64.(n_refs)
    check(ctx, ccall((:kgma_set_refs, libkgma), Cint,
        (Ptr{Cvoid}, Int32, Int32, Ptr{Float64}, Ptr{Int64}, Ptr{Float64}, Ptr{Int64}),
        ctx.h, k, m, refmat, Int64.(windowsizes), Float64.(thr_vec[1:m]), nref))
    g = genome_from_fasta(ctx, genome_path)
    try
        st = AlignState(g, consensus_seqs, 0,
            AffineGapScoreModel(EDNAFULL, gap_open = gap_open_score, gap_extend = gap_extend_score),
            get_aligns, align_vec)
        cb = align_hits ? @cfunction(align_trampoline, Cvoid,
            (Ptr{Cvoid}, Int32, Int32, Int64, Int64, Int64, Ptr{Int64}, Ptr{Int64})) : C_NULL
        GC.@preserve st begin
            check(ctx, ccall((:kgma_scan, libkgma), Cint,
                (Ptr{Cvoid}, Ptr{Cvoid}, Int32, Int64, Int64, UInt32, Ptr{Cvoid}, Ptr{Cvoid}),
                ctx.h, g.h, KGMA_MODE_OMN, buff, genome_pos, scan_flags(do_return_dists, float_chain),
                cb, pointer_from_objref(st)))
        end
        hits = fetch_hits(ctx)
        bodies = hit_bodies(g, hits)
        for (h, body) in zip(hits, bodies)
            seq_UnitRange = Int(h.lo):Int(h.hi)
            # record construction as in src/OmnGenomeMiner.jl:141-149
            push!(resultVec, FASTA.Record(
                identifier(g, h.contig) *
                    " | Dist = " * string(round(h.dist, digits = 2)) *
                    " | KFV = $(h.kfv)" *
                    " | MatchPos = $seq_UnitRange" *
                    " | GenomePos = $(h.genome_pos)" *
                    " | Len = " * string(last(seq_UnitRange) - first(seq_UnitRange) + 1),
                body))
            get_hit_loci && push!(hit_loci_vec, first(seq_UnitRange) + h.genome_pos)
        end
        if do_return_dists
            for j in 1:m; fetch_dists!(ctx, j, dist_vec_vec[j]) end
        end
    finally
        free!(g)
    end
    return nothing
end

export ac_gma_testing!, Omn_KmerGMA!, record_KmerGMA!, Context

end # module
