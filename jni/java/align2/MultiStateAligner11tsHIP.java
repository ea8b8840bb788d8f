package align2;

import java.nio.ByteBuffer;
import java.nio.ByteOrder;

/**
 * Batched affine-gap aligner backed by libbbmap_amd.so (MI355X) through libbbmap_amd_jni.so (jni/hip_glue.c).
 *
 * Where MultiStateAligner11tsJNI hands ONE fill per JNI call to native code and walks the returned matrix in Java
 * (fillLimitedXJNI + score2 + traceback2), this class collects the (read, site) pairs of a whole read list and runs them in one
 * call: each job is MSA.fillAndScoreLimited(read, ref, start, stop, minScore, gaps) [+ traceback], and comes back as the int[]
 * score vector and the match string the per-call methods return.  One instance per mapping thread, like every MSA.
 *
 * Typical use in BBMapThread.scoreSlow (which today calls msa.fillAndScoreLimited per site):
 * <pre>
 *   hip.clear();
 *   for (SiteScore ss : list) { ids[n++] = hip.add(bases, chacs, ss.start - pad, ss.stop + pad, minScore, ss.gaps, true); }
 *   hip.run();
 *   for (...) { int[] score = hip.score(id); byte[] match = hip.match(id); ... the unchanged per-site logic ... }
 * </pre>
 * Not compiled in the build image of the GPU library (it has no JDK); the native side is exercised through a mock JNIEnv
 * (jni/mock_jni_test.cpp, mode "glue").
 */
public final class MultiStateAligner11tsHIP implements AutoCloseable {

	static { System.loadLibrary("bbmap_amd_jni"); }

	/** job.flags of include/bbmap_amd.h */
	public static final int FILL_LIMITED_RAW = 0, FILL_UNLIMITED_RAW = 1, FILL_LIMITED = 2, CLAMP_WINDOW = 1 << 3, DO_SCORE = 1 << 4,
			DO_TRACEBACK = 1 << 5, NO_ITERATIONS = 1 << 6;
	public static final int FILL_AND_SCORE_LIMITED = FILL_LIMITED | CLAMP_WINDOW | DO_SCORE;
	public static final int SCHEME_11TS = 0, SCHEME_9PACBIO = 1;
	public static final int ST_OK = 0, ST_NULL = 1, ST_BAD_SHAPE = 2;

	private static final int JOB_BYTES = 40, RESULT_BYTES = 80, GAPS_BYTES = 68, MAX_GAPS = 16;

	private static native long create(int device, int maxRows, int maxColumns, int bandwidth, float bandwidthRatio, int scheme);
	private static native void destroy(long ctx);
	private static native void alignBatch(long ctx, int nJobs, ByteBuffer jobs, ByteBuffer reads, int readsBytes, ByteBuffer refs,
			int refsBytes, ByteBuffer results, ByteBuffer match, int matchStride);
	private static native void alignGappedBatch(long ctx, int nJobs, ByteBuffer jobs, ByteBuffer gaps, ByteBuffer reads, int readsBytes,
			ByteBuffer refs, int refsBytes, ByteBuffer results, ByteBuffer match, int matchStride);

	private long ctx;
	public final int maxRows, maxColumns;
	private final int matchStride;
	private int nJobs, readsBytes, refsBytes;
	private boolean anyGaps;
	private ByteBuffer jobs, gaps, reads, refs, results, match;

	/** bandwidth / bandwidthRatio: MSA.bandwidth, MSA.bandwidthRatio (static in the reference, MSA.java:864-865) */
	public MultiStateAligner11tsHIP(int device, int maxRows_, int maxColumns_, int bandwidth, float bandwidthRatio, int scheme) {
		maxRows = maxRows_;
		maxColumns = maxColumns_;
		ctx = create(device, maxRows, maxColumns, bandwidth, bandwidthRatio, scheme);
		// a traceback string never exceeds rows + columns symbols; gapped references add their gap symbols
		matchStride = ((maxRows + maxColumns + 2 + 128 * 24 + 15) / 16) * 16;
		grow(256, 256 * 160, 256 * 400);
	}

	private static ByteBuffer buf(int bytes) { return ByteBuffer.allocateDirect(bytes).order(ByteOrder.LITTLE_ENDIAN); }

	private void grow(int wantJobs, int wantReads, int wantRefs) {
		if (jobs == null || jobs.capacity() < wantJobs * JOB_BYTES) {
			final int n = Math.max(wantJobs, jobs == null ? 0 : 2 * (jobs.capacity() / JOB_BYTES));
			final ByteBuffer j = buf(n * JOB_BYTES), g = buf(n * GAPS_BYTES);
			if (jobs != null) { jobs.position(0).limit(nJobs * JOB_BYTES); j.put(jobs); gaps.position(0).limit(nJobs * GAPS_BYTES); g.put(gaps); }
			jobs = j; gaps = g;
			jobs.clear(); gaps.clear();
			results = buf(n * RESULT_BYTES);
			match = buf(n * matchStride);
		}
		if (reads == null || reads.capacity() < wantReads) {
			final ByteBuffer r = buf(Math.max(wantReads, reads == null ? 0 : 2 * reads.capacity()));
			if (reads != null) { reads.position(0).limit(readsBytes); r.put(reads); }
			reads = r; reads.clear();
		}
		if (refs == null || refs.capacity() < wantRefs) {
			final ByteBuffer r = buf(Math.max(wantRefs, refs == null ? 0 : 2 * refs.capacity()));
			if (refs != null) { refs.position(0).limit(refsBytes); r.put(refs); }
			refs = r; refs.clear();
		}
	}

	/** forget the previous batch */
	public void clear() { nJobs = 0; readsBytes = 0; refsBytes = 0; anyGaps = false; }

	/**
	 * Queue MSA.fillAndScoreLimited(read, ref, refStartLoc, refEndLoc, minScore, gaps) (MSA.java:103-134), plus traceback when asked.
	 * Only the window (clamped to the array, as the Java method clamps it) is copied; the job is rebased to it.
	 * @return the job's id for score() / match() after run()
	 */
	public int add(byte[] read, byte[] ref, int refStartLoc, int refEndLoc, int minScore, int[] gapArray, boolean traceback) {
		final int a = Math.max(0, refStartLoc), b = Math.min(ref.length - 1, refEndLoc);
		final int cols = Math.max(0, b - a + 1);
		grow(nJobs + 1, readsBytes + read.length, refsBytes + cols);
		reads.position(readsBytes); reads.put(read, 0, read.length);
		refs.position(refsBytes); if (cols > 0) { refs.put(ref, a, cols); }
		int flags = FILL_AND_SCORE_LIMITED | (traceback ? DO_TRACEBACK : 0);
		final int o = nJobs * JOB_BYTES;
		jobs.putLong(o, readsBytes); jobs.putLong(o + 8, refsBytes);
		jobs.putInt(o + 16, read.length); jobs.putInt(o + 20, cols);
		jobs.putInt(o + 24, 0); jobs.putInt(o + 28, cols - 1);
		jobs.putInt(o + 32, minScore); jobs.putInt(o + 36, flags);
		final int g = nJobs * GAPS_BYTES;
		final int ng = (gapArray == null) ? 0 : Math.min(gapArray.length, MAX_GAPS);
		gaps.putInt(g, ng);
		for (int i = 0; i < ng; i++) { gaps.putInt(g + 4 + 4 * i, gapArray[i] - a); }     // gap coordinates move with the window
		if (ng > 0) { anyGaps = true; }
		readsBytes += read.length; refsBytes += cols;
		return nJobs++;
	}

	/** one launch for everything queued since clear() */
	public void run() {
		if (nJobs == 0) { return; }
		if (anyGaps) { alignGappedBatch(ctx, nJobs, jobs, gaps, reads, readsBytes, refs, refsBytes, results, match, matchStride); }
		else { alignBatch(ctx, nJobs, jobs, reads, readsBytes, refs, refsBytes, results, match, matchStride); }
	}

	public int status(int job) { return results.getInt(job * RESULT_BYTES + 20); }
	/** cells the fill visited: what the native code adds to iterationsLimited / iterationsUnlimited */
	public long iterations(int job) { return results.getLong(job * RESULT_BYTES + 24); }

	/** what fillAndScoreLimited returns: {score, bestRefStart, bestRefStop} (+ maxRow, maxCol, maxState [, padLeft, padRight]), or null.
	 *  Reference positions are relative to the window handed to add(): add max(0, refStartLoc) to the two of them. */
	public int[] score(int job) {
		final int o = job * RESULT_BYTES, n = results.getInt(o + 64);
		if (n == 0) { return null; }
		final int[] s = new int[n];
		for (int i = 0; i < n; i++) { s[i] = results.getInt(o + 32 + 4 * i); }
		return s;
	}

	/** the traceback's match string (MSA.traceback), or null when none was asked for / the fill was null */
	public byte[] match(int job) {
		final int n = results.getInt(job * RESULT_BYTES + 68);
		if (n <= 0) { return null; }
		final byte[] m = new byte[n];
		match.position(job * matchStride); match.get(m, 0, n); match.clear();
		return m;
	}

	@Override
	public void close() { if (ctx != 0) { destroy(ctx); ctx = 0; } }
}
