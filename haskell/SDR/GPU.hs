{-# LANGUAGE ForeignFunctionInterface, RecordWildCards #-}

{-| GPU back-end for the FM hot path of the @sdr@ library, over the C ABI of @libsdr_hip.so@
    (@include/sdr_hip.h@, layer 3: Pipe operators).

    NOT COMPILED ANYWHERE IN THIS REPOSITORY'S BUILD IMAGE (no GHC there): this is the binding a
    maintainer of adamwalker/sdr would add next to SDR.Filter / SDR.Demod / SDR.Util.  The tested
    boundary is the C ABI itself (tests/test_abi.py, tests/test_gpu_pipes.py).

    Two levels.  (a) The reference's own records ('Filter' / 'Decimator' / 'Resampler', Filter.hs:116-144) with BOTH
    closures bound to device calls -- 'fastDecimatorCGpu', 'fastResamplerRGpu', 'fastFilterSymRGpu' -- so that the
    reference's unchanged 'firDecimator' / 'firResampler' / 'firFilter' drive the GPU:

    > decimator <- fastDecimatorCGpu 8 coeffsRFDecim          -- instead of fastDecimatorC info 8 coeffsRFDecim
    > ... firDecimator decimator samples ...                   -- fm.hs:36, unchanged

    (b) whole-Pipe operators that keep the stream state on the device.  They mirror the reference's:

    > firDecimatorGpu :: GpuDecimator -> Int -> Pipe (Vector (Complex Float)) (Vector (Complex Float)) IO ()   -- firDecimator, Filter.hs:574
    > firResamplerGpu :: GpuResampler -> Int -> Pipe (Vector Float) (Vector Float) IO ()                       -- firResampler, Filter.hs:679
    > firFilterGpu    :: GpuFilter    -> Int -> Pipe (Vector Float) (Vector Float) IO ()                       -- firFilter,    Filter.hs:532
    > fmDemodGpu      :: Pipe (Vector (Complex Float)) (Vector Float) IO ()                                    -- fmDemod,      Demod.hs:40
    > dcBlockingFilterGpu :: Pipe (Vector Float) (Vector Float) IO ()                                          -- dcBlockingFilter, Filter.hs:730
    > interleavedIQUnsignedByteToFloatGpu :: Vector CUChar -> Vector (Complex Float)                           -- Util.hs:137
    > fmReceiverGpu   :: GpuFmChain -> Int -> Int -> Pipe (Vector CUChar) (Vector Float) IO ()                 -- fm.hs:34-41 as one operator

    and produce the same vectors, bit for bit, as the AVX variants the reference selects on any AVX
    host (the One/Cross split at buffer seams included).
-}
module SDR.GPU (
    -- * The reference's own records, closures bound to the device (the reference's Pipes run unchanged)
    fastDecimatorCGpu, fastResamplerRGpu, fastFilterSymRGpu, fastFilterRGpu,
    -- * Device descriptors and whole-Pipe operators
    GpuDecimator, GpuResampler, GpuFilter,
    gpuDecimatorC, gpuResamplerR, gpuFilterSymR, gpuFilterR,
    firDecimatorGpu, firResamplerGpu, firFilterGpu, fmDemodGpu, dcBlockingFilterGpu,
    interleavedIQUnsignedByteToFloatGpu,
    GpuFmChain, gpuFmChain, fmReceiverGpu
    ) where

import           Control.Exception             (throwIO)
import           Control.Monad
import           Control.Concurrent.MVar (MVar, newMVar, withMVar)
import           Data.IORef
import           Control.Monad.Primitive       (RealWorld)
import           Data.Complex
import qualified Data.Vector.Storable.Mutable  as VSM
import           SDR.Filter                    (Filter (..), Decimator (..), Resampler (..))
import           Foreign
import           Foreign.C.String
import           Foreign.C.Types
import qualified Data.Vector.Storable         as VS
import           Pipes
import           System.IO.Unsafe             (unsafePerformIO)

data SdrDecimator
data SdrResampler
data SdrFilter
data SdrPipe
data SdrChain
data SdrStream

newtype GpuDecimator = GpuDecimator (Ptr SdrDecimator)
newtype GpuResampler = GpuResampler (Ptr SdrResampler)
newtype GpuFilter    = GpuFilter    (Ptr SdrFilter)
newtype GpuFmChain   = GpuFmChain   (Ptr SdrChain)

-- The imports are `safe`: the calls block on the device and must not stall the RTS.
foreign import ccall safe "sdrhip_last_error"          c_last_error        :: IO CString
foreign import ccall safe "sdrhip_decimator_create"    c_decimator_create  :: Ptr (Ptr SdrDecimator) -> CInt -> CInt -> CInt -> Ptr CFloat -> CInt -> IO CInt
foreign import ccall safe "sdrhip_resampler_create"    c_resampler_create  :: Ptr (Ptr SdrResampler) -> CInt -> CInt -> CInt -> CInt -> Ptr CFloat -> CInt -> IO CInt
foreign import ccall safe "sdrhip_filter_create"       c_filter_create     :: Ptr (Ptr SdrFilter) -> CInt -> CInt -> Ptr CFloat -> CInt -> IO CInt
foreign import ccall safe "sdrhip_filter_sym_create"   c_filter_sym_create :: Ptr (Ptr SdrFilter) -> CInt -> Ptr CFloat -> CInt -> IO CInt
foreign import ccall safe "sdrhip_pipe_fir_decimator"  c_pipe_decimator    :: Ptr (Ptr SdrPipe) -> Ptr SdrDecimator -> CInt -> IO CInt
foreign import ccall safe "sdrhip_pipe_fir_resampler"  c_pipe_resampler    :: Ptr (Ptr SdrPipe) -> Ptr SdrResampler -> CInt -> IO CInt
foreign import ccall safe "sdrhip_pipe_fir_filter"     c_pipe_filter       :: Ptr (Ptr SdrPipe) -> Ptr SdrFilter -> CInt -> IO CInt
foreign import ccall safe "sdrhip_pipe_fm_demod"       c_pipe_demod        :: Ptr (Ptr SdrPipe) -> IO CInt
foreign import ccall safe "sdrhip_pipe_push"           c_pipe_push         :: Ptr SdrPipe -> Ptr CFloat -> CInt -> IO CInt
foreign import ccall safe "sdrhip_pipe_pop"            c_pipe_pop          :: Ptr SdrPipe -> Ptr CFloat -> CInt -> IO CInt
foreign import ccall safe "sdrhip_pipe_dc_blocker"     c_pipe_dc_blocker   :: Ptr (Ptr SdrPipe) -> IO CInt
-- throughput knob: stage equal-sized pushes and submit them n at a time (same output blocks, n blocks of latency)
foreign import ccall safe "sdrhip_pipe_set_coalesce"   c_pipe_set_coalesce :: Ptr SdrPipe -> CInt -> IO CInt
foreign import ccall safe "sdrhip_fm_stream_set_coalesce" c_stream_set_coalesce :: Ptr SdrStream -> CInt -> IO CInt
-- adaptive submission (on by default: pushes that arrive while the GPU is busy share a launch); 0 switches it off
foreign import ccall safe "sdrhip_pipe_set_adaptive"   c_pipe_set_adaptive :: Ptr SdrPipe -> CInt -> IO CInt
foreign import ccall safe "sdrhip_fm_stream_set_adaptive" c_stream_set_adaptive :: Ptr SdrStream -> CInt -> IO CInt
-- blocks ready to pop after collecting (without waiting) what the GPU has finished: for consumers that want the audio of the
-- block they just pushed before the next one arrives
-- (`safe`: collecting a finished batch may wait on a HIP event and copies whole result batches; an `unsafe` call would hold
-- the capability and the GC for that long)
foreign import ccall safe "sdrhip_pipe_poll"      c_pipe_poll   :: Ptr SdrPipe -> IO CInt
foreign import ccall safe "sdrhip_fm_stream_poll" c_stream_poll :: Ptr SdrStream -> IO CInt
foreign import ccall safe "sdrhip_fm_chain_create"     c_chain_create      :: Ptr (Ptr SdrChain) -> CInt -> CInt -> Ptr CFloat -> CInt -> CInt -> CInt -> Ptr CFloat -> CInt -> Ptr CFloat -> CInt -> CFloat -> Int64 -> IO CInt
foreign import ccall safe "sdrhip_fm_stream_create"    c_stream_create     :: Ptr (Ptr SdrStream) -> Ptr SdrChain -> CInt -> CInt -> IO CInt
foreign import ccall safe "sdrhip_fm_stream_push"      c_stream_push       :: Ptr SdrStream -> Ptr CUChar -> CInt -> IO CInt
foreign import ccall safe "sdrhip_fm_stream_pop"       c_stream_pop        :: Ptr SdrStream -> Ptr CFloat -> CInt -> IO CInt
foreign import ccall safe "convertCAVX"                c_convertCAVX       :: CInt -> Ptr CUChar -> Ptr CFloat -> IO ()
-- the drop-in symbols return void: a failure inside one reaches this handler instead of abort() (sdrhip_set_error_handler)
type ErrorHandler = CInt -> CString -> IO ()
foreign import ccall "wrapper" mkErrorHandler :: ErrorHandler -> IO (FunPtr ErrorHandler)
foreign import ccall safe "sdrhip_set_error_handler"   c_set_error_handler :: FunPtr ErrorHandler -> IO ()
-- the record seam (include/sdr_hip.h, "the record seam on HOST vectors"): One = the C SIMD kernel on one buffer,
-- Cross = the sequential kernel on `drop i last ++ next` (FilterInternal.hs:397-423)
foreign import ccall safe "sdrhip_filter_num_coeffs"    c_filter_num_coeffs    :: Ptr SdrFilter -> IO CInt
foreign import ccall safe "sdrhip_decimator_num_coeffs" c_decimator_num_coeffs :: Ptr SdrDecimator -> IO CInt
foreign import ccall safe "sdrhip_resampler_num_coeffs" c_resampler_num_coeffs :: Ptr SdrResampler -> IO CInt
foreign import ccall safe "sdrhip_filter_one"       c_filter_one       :: Ptr SdrFilter -> CInt -> Ptr CFloat -> Ptr CFloat -> IO CInt
foreign import ccall safe "sdrhip_filter_cross"     c_filter_cross     :: Ptr SdrFilter -> CInt -> Ptr CFloat -> CInt -> Ptr CFloat -> CInt -> Ptr CFloat -> IO CInt
foreign import ccall safe "sdrhip_decimator_one"    c_decimator_one    :: Ptr SdrDecimator -> CInt -> Ptr CFloat -> Ptr CFloat -> IO CInt
foreign import ccall safe "sdrhip_decimator_cross"  c_decimator_cross  :: Ptr SdrDecimator -> CInt -> Ptr CFloat -> CInt -> Ptr CFloat -> CInt -> Ptr CFloat -> IO CInt
foreign import ccall safe "sdrhip_resampler_one"    c_resampler_one    :: Ptr SdrResampler -> CInt -> CInt -> Ptr CFloat -> CInt -> Ptr CFloat -> IO CInt
foreign import ccall safe "sdrhip_resampler_cross"  c_resampler_cross  :: Ptr SdrResampler -> CInt -> CInt -> Ptr CFloat -> CInt -> Ptr CFloat -> CInt -> Ptr CFloat -> IO CInt

-- | SDRHIP_ORDER_AVX: reproduce the variant 'SDR.CPUID.featureSelect' picks on any AVX host.
orderAVX :: CInt
orderAVX = 2

-- | The reference's hand-rolled @assert@ calls @error@ (Filter.hs:526-527); so does a negative status.
check :: CInt -> IO CInt
check rc
    | rc < 0    = c_last_error >>= peekCString >>= \msg -> error ("sdr_hip: " ++ msg)
    | otherwise = return rc

-- | Failures of the void drop-in symbols ('convertCAVX' and the @foreign import@s of SDR.FilterInternal when the
--   reference is relinked against libsdr_hip.so): the C side records the message, calls this handler and returns to
--   its caller; the wrapper raises after the foreign call -- the reference raises from its Pipes too (Filter.hs:526-527),
--   it does not die.  Installed once, on first use of this module.
{-# NOINLINE dropInFailure #-}
dropInFailure :: IORef (Maybe String)
dropInFailure = unsafePerformIO $ do
    ref <- newIORef Nothing
    h <- mkErrorHandler $ \_code msg -> peekCString msg >>= writeIORef ref . Just
    c_set_error_handler h
    return ref

-- | One drop-in call at a time: the failure flag above is process-global (the C handler has no per-call argument), so with
--   several Haskell threads calling drop-in symbols one thread's failure could be cleared or claimed by another.  The
--   reference's pipeline is a single thread (fm.hs:30-41); for anything else the lock makes the flag per call.
{-# NOINLINE dropInLock #-}
dropInLock :: MVar ()
dropInLock = unsafePerformIO (newMVar ())

-- | Run a void drop-in call and raise if it reported a failure.
dropIn :: IO () -> IO ()
dropIn act = withMVar dropInLock $ \_ -> do
    writeIORef dropInFailure Nothing
    act
    readIORef dropInFailure >>= maybe (return ()) (\msg -> throwIO (userError ("sdr_hip: " ++ msg)))

withCoeffs :: [Float] -> (Ptr CFloat -> CInt -> IO a) -> IO a
withCoeffs cs act = withArrayLen (map realToFrac cs) $ \n p -> act p (fromIntegral n)

-- | 'fastDecimatorC' (Filter.hs:352-356) on the GPU: complex data, real taps.
gpuDecimatorC :: Int -> [Float] -> IO GpuDecimator
gpuDecimatorC factor coeffs = alloca $ \pp -> do
    _ <- withCoeffs coeffs $ \p n -> c_decimator_create pp orderAVX 1 (fromIntegral factor) p n >>= check
    GpuDecimator <$> peek pp

-- | 'fastResamplerR' (Filter.hs:468-473) on the GPU.
gpuResamplerR :: Int -> Int -> [Float] -> IO GpuResampler
gpuResamplerR interp decim coeffs = alloca $ \pp -> do
    _ <- withCoeffs coeffs $ \p n -> c_resampler_create pp orderAVX 0 (fromIntegral interp) (fromIntegral decim) p n >>= check
    GpuResampler <$> peek pp

-- | 'fastFilterSymR' (Filter.hs:258-261) on the GPU: pass the FIRST HALF of an even-length linear-phase filter.
gpuFilterSymR :: [Float] -> IO GpuFilter
gpuFilterSymR half = alloca $ \pp -> do
    _ <- withCoeffs half $ \p n -> c_filter_sym_create pp orderAVX p n >>= check
    GpuFilter <$> peek pp

-- | 'fastFilterR' (Filter.hs:191-194) on the GPU.
gpuFilterR :: [Float] -> IO GpuFilter
gpuFilterR coeffs = alloca $ \pp -> do
    _ <- withCoeffs coeffs $ \p n -> c_filter_create pp orderAVX 0 p n >>= check
    GpuFilter <$> peek pp

-- ---------------------------------------------------------------------------------------------
-- The reference's records (Filter.hs:116-144) with both closures bound to the device.  These are
-- drop-in replacements for 'fastDecimatorC' / 'fastResamplerR' / 'fastFilterSymR' / 'fastFilterR':
-- hand the result to the reference's OWN 'firDecimator' / 'firResampler' / 'firFilter' (Filter.hs:532-727)
-- and the pipeline of examples/fm/fm.hs runs unchanged, every vector identical bit for bit.
-- (One FFI round trip per closure call; the whole-Pipe operators below amortise it better.)
-- ---------------------------------------------------------------------------------------------
withIn :: Storable a => VS.Vector a -> (Ptr CFloat -> CInt -> IO b) -> IO b
withIn v act = VS.unsafeWith (VS.unsafeCast v) $ \p -> act p (fromIntegral (VS.length v))

withOut :: Storable a => VSM.MVector RealWorld a -> (Ptr CFloat -> IO b) -> IO b
withOut v act = VSM.unsafeWith (VSM.unsafeCast v) act

-- | 'fastDecimatorC' (Filter.hs:352-356): @Decimator IO Vector MVector (Complex Float)@.
fastDecimatorCGpu :: Int -> [Float] -> IO (Decimator IO VS.Vector VSM.MVector (Complex Float))
fastDecimatorCGpu decimationD coeffs = do
    GpuDecimator d <- gpuDecimatorC decimationD coeffs
    numCoeffsD <- fromIntegral <$> c_decimator_num_coeffs d
    let decimateOne num inBuf outBuf =
            void $ withIn inBuf $ \i _ -> withOut outBuf $ \o -> c_decimator_one d (fromIntegral num) i o >>= check
        decimateCross num lastBuf nextBuf outBuf =
            void $ withIn lastBuf $ \l nl -> withIn nextBuf $ \n nn -> withOut outBuf $ \o ->
                c_decimator_cross d (fromIntegral num) l nl n nn o >>= check
    return Decimator {..}

-- | 'fastFilterSymR' (Filter.hs:258-261): pass the first half of an even-length linear-phase filter.
fastFilterSymRGpu :: [Float] -> IO (Filter IO VS.Vector VSM.MVector Float)
fastFilterSymRGpu half = gpuFilterSymR half >>= filterRecord

-- | 'fastFilterR' (Filter.hs:191-194).
fastFilterRGpu :: [Float] -> IO (Filter IO VS.Vector VSM.MVector Float)
fastFilterRGpu coeffs = gpuFilterR coeffs >>= filterRecord

filterRecord :: GpuFilter -> IO (Filter IO VS.Vector VSM.MVector Float)
filterRecord (GpuFilter f) = do
    numCoeffsF <- fromIntegral <$> c_filter_num_coeffs f
    let filterOne num inBuf outBuf =
            void $ withIn inBuf $ \i _ -> withOut outBuf $ \o -> c_filter_one f (fromIntegral num) i o >>= check
        filterCross num lastBuf nextBuf outBuf =
            void $ withIn lastBuf $ \l nl -> withIn nextBuf $ \n nn -> withOut outBuf $ \o ->
                c_filter_cross f (fromIntegral num) l nl n nn o >>= check
    return Filter {..}

-- | 'fastResamplerR' (Filter.hs:468-473).  The state carried between calls is the reference's own
--   @(group, offset)@ (mkResampler, Filter.hs:408-425): 'resampleOne' starts in polyphase group @group@
--   and learns the next group from the C call; 'resampleCross' starts at filter offset @offset@ and
--   learns the next offset from it.
fastResamplerRGpu :: Int -> Int -> [Float] -> IO (Resampler IO VS.Vector VSM.MVector Float)
fastResamplerRGpu interpolationR decimationR coeffs = do
    GpuResampler r <- gpuResamplerR interpolationR decimationR coeffs
    numCoeffsR <- fromIntegral <$> c_resampler_num_coeffs r
    let offsetOf group = interpolationR - 1 - ((interpolationR + group * decimationR - 1) `mod` interpolationR)
        resampleOne (group, _) num inBuf outBuf = do
            group' <- fmap fromIntegral $ withIn inBuf $ \i ni -> withOut outBuf $ \o ->
                          c_resampler_one r (fromIntegral group) (fromIntegral num) i ni o >>= check
            let offset' = offsetOf group'
            return ((group', offset'), offset')
        resampleCross (group, offset) num lastBuf nextBuf outBuf = do
            offset' <- fmap fromIntegral $ withIn lastBuf $ \l nl -> withIn nextBuf $ \n nn -> withOut outBuf $ \o ->
                           c_resampler_cross r (fromIntegral offset) (fromIntegral num) l nl n nn o >>= check
            return (((group + num) `mod` interpolationR, offset'), offset')
        startDat = (0, 0) :: (Int, Int)
    return Resampler {..}

-- | Forward blocks through one C pipe.  @wIn@ / @wOut@: floats per element (2 for complex).
--   Output blocks have exactly @blockSizeOut@ elements (advanceOutBuf, Filter.hs:516-523).
runPipe :: (Storable a, Storable b) => Int -> Int -> Ptr SdrPipe -> Int -> Pipe (VS.Vector a) (VS.Vector b) IO ()
runPipe wIn wOut pipe blockSizeOut = forever $ do
    inp   <- await
    ready <- lift $ VS.unsafeWith (VS.unsafeCast inp) $ \ptr ->
                 c_pipe_push pipe ptr (fromIntegral (VS.length inp)) >>= check
    replicateM_ (fromIntegral ready) $ do
        out <- lift $ do
            fp <- mallocForeignPtrArray (wOut * blockSizeOut) :: IO (ForeignPtr CFloat)
            _  <- withForeignPtr fp $ \o -> c_pipe_pop pipe o (fromIntegral blockSizeOut) >>= check
            return $ VS.unsafeCast $ VS.unsafeFromForeignPtr0 fp (wOut * blockSizeOut)
        yield out
  where _ = wIn

mkPipe :: (Ptr (Ptr SdrPipe) -> IO CInt) -> IO (Ptr SdrPipe)
mkPipe create = alloca $ \pp -> create pp >>= check >> peek pp

firDecimatorGpu :: GpuDecimator -> Int -> Pipe (VS.Vector (Complex Float)) (VS.Vector (Complex Float)) IO ()
firDecimatorGpu (GpuDecimator d) blockSizeOut = do
    pipe <- lift $ mkPipe $ \pp -> c_pipe_decimator pp d (fromIntegral blockSizeOut)
    runPipe 2 2 pipe blockSizeOut

firResamplerGpu :: GpuResampler -> Int -> Pipe (VS.Vector Float) (VS.Vector Float) IO ()
firResamplerGpu (GpuResampler r) blockSizeOut = do
    pipe <- lift $ mkPipe $ \pp -> c_pipe_resampler pp r (fromIntegral blockSizeOut)
    runPipe 1 1 pipe blockSizeOut

firFilterGpu :: GpuFilter -> Int -> Pipe (VS.Vector Float) (VS.Vector Float) IO ()
firFilterGpu (GpuFilter f) blockSizeOut = do
    pipe <- lift $ mkPipe $ \pp -> c_pipe_filter pp f (fromIntegral blockSizeOut)
    runPipe 1 1 pipe blockSizeOut

-- | 'fmDemod' (Demod.hs:40-46): one output vector per input vector; the block size passed to
--   'runPipe' is only an upper bound here, the C side returns each block's own length.
fmDemodGpu :: Pipe (VS.Vector (Complex Float)) (VS.Vector Float) IO ()
fmDemodGpu = do
    pipe <- lift $ mkPipe c_pipe_demod
    forever $ do
        inp   <- await
        ready <- lift $ VS.unsafeWith (VS.unsafeCast inp) $ \ptr ->
                     c_pipe_push pipe ptr (fromIntegral (VS.length inp)) >>= check
        replicateM_ (fromIntegral ready) $ do
            out <- lift $ do
                let cap = VS.length inp
                fp  <- mallocForeignPtrArray cap :: IO (ForeignPtr CFloat)
                n   <- withForeignPtr fp $ \o -> c_pipe_pop pipe o (fromIntegral cap) >>= check
                return $ VS.unsafeCast $ VS.unsafeFromForeignPtr0 fp (fromIntegral n)
            yield out

-- | 'interleavedIQUnsignedByteToFloatFast' (Util.hs:137-138) through the drop-in symbol.
interleavedIQUnsignedByteToFloatGpu :: VS.Vector CUChar -> VS.Vector (Complex Float)
interleavedIQUnsignedByteToFloatGpu inBuf = unsafePerformIO $ do
    fp <- mallocForeignPtrArray (VS.length inBuf) :: IO (ForeignPtr CFloat)
    VS.unsafeWith inBuf $ \iPtr -> withForeignPtr fp $ \oPtr ->
        dropIn $ c_convertCAVX (fromIntegral $ VS.length inBuf) iPtr oPtr
    return $ VS.unsafeCast $ VS.unsafeFromForeignPtr0 fp (VS.length inBuf)

-- | 'dcBlockingFilter' (Filter.hs:730-739): one output vector per input vector, the filter state
--   carried on the device.  Worth it for long vectors only (a short block is one dependent chain).
dcBlockingFilterGpu :: Pipe (VS.Vector Float) (VS.Vector Float) IO ()
dcBlockingFilterGpu = do
    pipe <- lift $ mkPipe c_pipe_dc_blocker
    forever $ do
        inp   <- await
        ready <- lift $ VS.unsafeWith (VS.unsafeCast inp) $ \ptr ->
                     c_pipe_push pipe ptr (fromIntegral (VS.length inp)) >>= check
        replicateM_ (fromIntegral ready) $ do
            out <- lift $ do
                let cap = VS.length inp
                fp  <- mallocForeignPtrArray cap :: IO (ForeignPtr CFloat)
                n   <- withForeignPtr fp $ \o -> c_pipe_pop pipe o (fromIntegral cap) >>= check
                return $ VS.unsafeCast $ VS.unsafeFromForeignPtr0 fp (fromIntegral n)
            yield out

-- | The whole receiver of examples/fm/fm.hs:34-41 -- convert, decimate, demodulate, resample, filter,
--   gain -- with every intermediate resident in device memory.  Arguments as in fm.hs: decimation and
--   its taps, interpolation / decimation and the resampler taps, the HALF taps of the symmetric audio
--   filter, the gain ('P.map (VG.map (* 0.2))'), and the source block size ('samples'; the seams of the
--   reference's Pipes fall at multiples of it).
gpuFmChain :: Int -> [Float] -> Int -> Int -> [Float] -> [Float] -> Float -> Int -> IO GpuFmChain
gpuFmChain decimation rfTaps interpolation decimation2 resampTaps audioHalf gain block =
    withArrayLen (map realToFrac rfTaps) $ \n1 p1 ->
    withArrayLen (map realToFrac resampTaps) $ \n2 p2 ->
    withArrayLen (map realToFrac audioHalf) $ \n3 p3 ->
    alloca $ \pp -> do
        _ <- c_chain_create pp orderAVX (fromIntegral decimation) p1 (fromIntegral n1)
                 (fromIntegral interpolation) (fromIntegral decimation2) p2 (fromIntegral n2)
                 p3 (fromIntegral n3) (realToFrac gain) (fromIntegral block) >>= check
        GpuFmChain <$> peek pp

-- | u8 IQ blocks from 'sdrStream' in, audio blocks of exactly @blockSizeOut@ floats out: same blocks,
--   bit for bit, as the five stages it replaces.  @maxBlock@ = the largest source block (a multiple of
--   the chain's block size) that will be pushed.
fmReceiverGpu :: GpuFmChain -> Int -> Int -> Pipe (VS.Vector CUChar) (VS.Vector Float) IO ()
fmReceiverGpu (GpuFmChain c) maxBlock blockSizeOut = do
    st <- lift $ alloca $ \pp -> c_stream_create pp c (fromIntegral maxBlock) (fromIntegral blockSizeOut) >>= check >> peek pp
    forever $ do
        inp   <- await
        ready <- lift $ VS.unsafeWith inp $ \ptr -> c_stream_push st ptr (fromIntegral (VS.length inp `div` 2)) >>= check
        replicateM_ (fromIntegral ready) $ do
            out <- lift $ do
                fp <- mallocForeignPtrArray blockSizeOut :: IO (ForeignPtr CFloat)
                _  <- withForeignPtr fp $ \o -> c_stream_pop st o (fromIntegral blockSizeOut) >>= check
                return $ VS.unsafeCast $ VS.unsafeFromForeignPtr0 fp blockSizeOut
            yield out
